// Warp-specialised persistent bf16 MFMA GEMM for the forward / input-gradient launches of the Linear layers (models/dit.py:118-155).
//
// What round 4's measurements say paces the other two kernels' K loops: neither bytes in flight (gemm_pd_kernel stages 1.5 x the
// bytes per MFMA and runs within 10 %) nor barriers (one per K tile instead of four: +-3 %), but the ISSUE of the LDS-DMA
// instructions by the waves that also issue the MFMAs.  A `buffer_load ... lds` holds its wave for ~100 cycles in a loaded phase
// (MI355X_MICROARCH.md, "LDS-DMA piece issue cost"), the wave is in-order, and K tile time = MFMA time + DMA issue time fits both
// kernels (256-row: 2 waves x (1024 + 8 x 100) = 3650 cycles per SIMD and K tile against 3700 measured; 128-row: 2 x (384 + 5 x 100)
// = 1770 against 1620).  Here the two jobs are given to different waves, which the SIMD issues from independently:
//   * 12 waves per workgroup, one workgroup per CU: waves 0-7 CONSUME (2 (M) x 4 (N), wave tile 64 x 16 NTW, 64 accumulator
//     registers; fragment reads + MFMAs + epilogue, never a vector-memory instruction inside the K loop), waves 8-11 LOAD (all
//     LDS-DMA of the workgroup: 2 x (2 + NTW) pieces of 1 KiB per K tile each, two K tiles ahead, counted vmcnt); 3 waves per
//     SIMD fit because a 64 x 64 wave tile needs < 168 registers;
//   * tile 128 x (256 | 192), K in 64-deep tiles through a ring of THREE stages; ONE workgroup barrier per K tile -- it makes the
//     K tile's pieces visible to the consumers and tells the loaders that everybody has left the stage the next pieces go into;
//   * the epilogue (gemm_epi.h kinds, the 256-row kernel's pipelined wave-private form: accumulators -> f32 LDS image -> 8 columns
//     per lane, 16-byte accesses) runs in the ring stage that is free at an item's end, behind one extra barrier per item; the
//     loaders meanwhile fetch the next item's first two K tiles.
// Layouts: A k-major; B k-major (forward) or mn-major (input gradients).  No K split, no grouped mode.
#pragma once
#include "gemm_p8_kernel.h"

#define WS_BM 128
// measurement (make exp XF=-DWS_KROT=1): every workgroup of an XCD walks K from a different starting tile (wrapping round), so
// that the CUs sharing an L2 do not all ask for the same operand lines at the same moment
#ifndef WS_KROT
#define WS_KROT 0
#endif
template <int NTW> struct WsCfg {
    static constexpr int BN = 64 * NTW, WN = 16 * NTW;
    static constexpr int LS = 2 + NTW;                       // 8 KiB parts per stage: A parts 0, 1 | B parts 0 .. NTW-1
    static constexpr int a_bytes = 2 * P8_PART;
    static constexpr int stage_bytes = LS * P8_PART;
    static constexpr int lds_bytes = 3 * stage_bytes;
    static constexpr int PPS = 2 * LS;                       // DMA instructions per loader wave and K tile
};

template <int N> __device__ __forceinline__ void ws_vmwait() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <bool BKM, int NTW, int EPI, int NLW = 4>      // NLW loader waves: 4 (12 waves, <= 168 registers) or 8 (16 waves, <= 128 registers: NTW = 3 only)
__global__ void __launch_bounds__(512 + 64 * NLW)
gemm_ws_kernel(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ B, int64_t ldb, int nk, int tiles_m, int tiles_n,
               EpiDev e) {
    using Cfg = WsCfg<NTW>;
    using EK = EpiKind<EPI>;
    constexpr int WN = Cfg::WN;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [3 stages][A parts 0-1 | B parts]
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int G = gridDim.x, n_items = tiles_m * tiles_n;
    int it0;
    {
        const int b = blockIdx.x, x = b & 7, q = G >> 3, r = G & 7;
        it0 = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
    }
    auto decode = [&](int it, int& tm, int& tn) __attribute__((always_inline)) {
        if (tiles_n >= 8 && e.debug != 3) {      // 8 x 4 blocks of tiles per XCD run (see gemm_pd_kernel.h)
            const int gsz = 8 * tiles_n, gi = it / gsz, within = it - gi * gsz;
            const int rows = tiles_m - 8 * gi < 8 ? tiles_m - 8 * gi : 8;
            tn = within / rows;
            tm = 8 * gi + within - tn * rows;
        } else {
            tm = it / tiles_n;
            tn = it - tm * tiles_n;
        }
    };

    if (wid >= 8) {
        // ================================ loader waves ================================
        const int lw = wid - 8;
        int iss_item = it0, iss_kt = 0, iss_stage = 0;
        __amdgpu_buffer_rsrc_t rs_a = epi_rsrc(A), rs_b = epi_rsrc(B);
        unsigned so_a = 0, so_b = 0;
        const unsigned step_a = 128u, step_b = BKM ? 128u : (unsigned)(64 * ldb * 2);
        constexpr int NH = 8 / NLW;              // eighths of a part per loader wave
        unsigned off_a[2][2], off_b[NTW][2];
        const int krot = WS_KROT ? (int)((blockIdx.x >> 3) * 5u % (unsigned)nk) : 0;
        int kpos = 0;
        auto open = [&]() __attribute__((always_inline)) {
            kpos = krot;
            so_a = (unsigned)kpos * step_a;
            so_b = (unsigned)kpos * step_b;
            iss_kt = 0;
            if (iss_item >= n_items) {           // past the last item: out-of-range pieces (zeros into a stage nobody reads)
#pragma unroll
                for (int h = 0; h < NH; ++h) {
#pragma unroll
                    for (int p = 0; p < 2; ++p) off_a[p][h] = EPI_OOB;
#pragma unroll
                    for (int p = 0; p < NTW; ++p) off_b[p][h] = EPI_OOB;
                }
                return;
            }
            int tm, tn;
            decode(iss_item, tm, tn);
            const int64_t m0 = (int64_t)tm * WS_BM, n0 = (int64_t)tn * Cfg::BN;
            const int mvalid = e.M - m0 < WS_BM ? (int)(e.M - m0) : WS_BM;
            const int nvalid = e.N - n0 < Cfg::BN ? (int)(e.N - n0) : Cfg::BN;
            rs_a = epi_rsrc(A + m0 * lda);
            rs_b = epi_rsrc(BKM ? B + n0 * ldb : B + n0);
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                const int w8 = NH * lw + h;      // which eighth of a part (8 rows of 128 bytes) this instruction fills
#pragma unroll
                for (int p = 0; p < 2; ++p) {    // A part p: rows {32 p ..} and {64 + 32 p ..} of the 128-row tile
                    const int r = 8 * w8 + (lane >> 3);
                    const int chunk = (lane & 7) ^ ((r >> 1) & 7);
                    int R = r < 32 ? 32 * p + r : 32 + 32 * p + r;
                    R = R < mvalid ? R : mvalid - 1;
                    off_a[p][h] = (unsigned)(R * lda * 2) + chunk * 16;
                }
#pragma unroll
                for (int p = 0; p < NTW; ++p) off_b[p][h] = p8_src_off<BKM>(false, p, w8, lane, ldb, nvalid);
            }
        };
        auto issue_set = [&]() __attribute__((always_inline)) {
            char* dst = smem + iss_stage * Cfg::stage_bytes;
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                const int w8 = NH * lw + h;
#pragma unroll
                for (int p = 0; p < NTW; ++p) p8_dma16(rs_b, dst + Cfg::a_bytes + p * P8_PART + w8 * 1024, off_b[p][h], so_b);
#pragma unroll
                for (int p = 0; p < 2; ++p) p8_dma16(rs_a, dst + p * P8_PART + w8 * 1024, off_a[p][h], so_a);
            }
            iss_stage = iss_stage == 2 ? 0 : iss_stage + 1;
            so_a += step_a;
            so_b += step_b;
            if (WS_KROT && ++kpos == nk) { kpos = 0; so_a = so_b = 0; }
            if (++iss_kt == nk) {
                iss_item += G;
                open();
            }
        };
        open();
        issue_set();
        issue_set();
        for (int it = it0; it < n_items; it += G) {
            for (int kt = 0; kt < nk; ++kt) {
                ws_vmwait<Cfg::PPS * NH / 2>();  // this wave's pieces of the K tile about to be consumed (younger: the next set)
                __builtin_amdgcn_s_barrier();    // ... visible to the consumers; everybody has left the stage of two K tiles ago
                issue_set();                     // the set two K tiles ahead goes there
            }
            __builtin_amdgcn_s_barrier();        // (the consumers' barrier in front of their epilogue)
        }
        ws_vmwait<0>();                          // the trailing out-of-range pieces still target this workgroup's LDS
        return;
    }

    // ================================ consumer waves ================================
    const int wr = wid >> 2, wc = wid & 3;
    const int wn0 = wc * WN;
    int cstage = 0;
    for (int it = it0; it < n_items; it += G) {
        int tm, tn;
        decode(it, tm, tn);
        const int64_t m0 = (int64_t)tm * WS_BM, n0 = (int64_t)tn * Cfg::BN;
        f32x4 acc[4][NTW];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int u = 0; u < NTW; ++u) acc[i][u] = f32x4{0, 0, 0, 0};
        int lane_k = lane;
        asm volatile("" : "+v"(lane_k));
        for (int kt = 0; kt < nk; ++kt) {
            __builtin_amdgcn_s_barrier();
            const char* st = smem + cstage * Cfg::stage_bytes;
            cstage = cstage == 2 ? 0 : cstage + 1;
            bf16x8 bfr[2][NTW], af[2][4];
#pragma unroll
            for (int u = 0; u < NTW; ++u) {
                const int n = wn0 + 16 * u;
#pragma unroll
                for (int s = 0; s < 2; ++s) bfr[s][u] = p8_frag<BKM>(st + Cfg::a_bytes + (n >> 6) * P8_PART, n & 63, s, lane_k);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int s = 0; s < 2; ++s) af[s][2 * j + t] = p8_frag<true>(st + j * P8_PART, wr * 32 + 16 * t, s, lane_k);
            if (!BKM) {             // transposed fragments are read by inline asm: the compiler does not wait for them itself
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int u = 0; u < NTW; ++u) acc[i][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[s][u], af[s][i], acc[i][u], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        }
        __builtin_amdgcn_s_barrier();            // every consumer has read its last fragments: the stage just consumed is the epilogue's

        // ---- epilogue: 4 row tiles of 16 rows through this wave's private 4 KiB f32 image in the free stage ----
        EpiDev ei = e;
        const int free_stage = cstage == 0 ? 2 : cstage - 1;
        char* ep = smem + free_stage * Cfg::stage_bytes + wid * 4096;
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        const int wr_row = lane_e & 15, wr_g = lane_e >> 4;
        const int rd_row = lane_e >> 3, rd_c8 = lane_e & 7;
        const unsigned ep_base = (unsigned)(uintptr_t)(lds_ptr_t)ep;
        const unsigned ep_w = ep_base + wr_row * 256 + ((wr_g ^ (wr_row & 3)) << 4);
        unsigned ep_r[2];
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int rr = pass * 8 + rd_row;
            ep_r[pass] = ep_base + rr * 256 + (((2 * rd_c8) ^ rr) << 4);
        }
        const int64_t n = n0 + wn0 + 8 * rd_c8;
        const bool col_ok = 8 * rd_c8 < WN && n < ei.N;
        const int64_t n_ld = col_ok ? n : n0;
        const int64_t m_last = ei.M - 1;
        f32x4 b0 = {0, 0, 0, 0}, b1 = {0, 0, 0, 0};
        if (EPI != P8_DGELU && ei.bias) {
            b0 = load4(ei.bias + n_ld);
            b1 = load4(ei.bias + n_ld + 4);
            asm volatile("" ::"v"(b0), "v"(b1));
        }
        f32x4 s0 = {0, 0, 0, 0}, s1 = {0, 0, 0, 0};
        float qmax = 0.f;
        const int loc_col = wn0 + 8 * rd_c8;
        const bool c_f32 = EK::out_f32(ei);
        const int64_t tile_off = m0 * ei.ldc + n0;
        const __amdgpu_buffer_rsrc_t rs_c = c_f32 ? epi_rsrc((const float*)ei.C + tile_off) : epi_rsrc((const bf16_t*)ei.C + tile_off);
        const __amdgpu_buffer_rsrc_t rs_aux = epi_rsrc(EK::aux_out(ei) ? (const void*)((const bf16_t*)ei.aux_out + tile_off) : (const void*)ei.C);
        const int64_t mrow0 = m0 + wr * 64 + rd_row;
        const unsigned rpb = (unsigned)ei.rpb;
        unsigned smp = 0, rin = 0;
        constexpr int PD = EPI == P8_DGELU ? 3 : 2;
        EpiOps ops[PD + 1];
        const bool with_ops = EK::loads && (EK::act2(ei) || EK::gate(ei) || EK::resid(ei) || EK::rowadd(ei));
        int64_t m_ld = mrow0;
        auto load_next = [&](EpiOps& dst, bool first) __attribute__((always_inline)) {
            if (first) {
                const int64_t mc = m_ld < m_last ? m_ld : m_last;
                smp = (unsigned)mc / rpb;
                rin = (unsigned)mc % rpb;
                epi_load8<EPI>(ei, (unsigned)mc, n_ld, smp, rin, dst);
                return;
            }
            int64_t mn = m_ld + 8;
            m_ld = mn;
            if (mn <= m_last) {
                rin += 8;
                if (rin >= rpb) {
                    if (rpb >= 8) { rin -= rpb; smp += 1; }
                    else { smp += rin / rpb; rin %= rpb; }
                }
            } else {
                mn = m_last;
                smp = (unsigned)mn / rpb;
                rin = (unsigned)mn % rpb;
            }
            epi_load8<EPI>(ei, (unsigned)mn, n_ld, smp, rin, dst);
        };
        if (with_ops) {
#pragma unroll
            for (int d = 0; d < PD; ++d) load_next(ops[d], d == 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int u = 0; u < NTW; ++u)
                asm volatile("ds_write_b128 %0, %1" ::"v"(ep_w + (unsigned)(((4 * u) ^ (wr_row & 12)) << 4)), "v"(acc[i][u]) : "memory");
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                const int k = 2 * i + pass;
                const int64_t m = mrow0 + 8 * k;
                if (with_ops && k + PD < 8) load_next(ops[(k + PD) % (PD + 1)], false);
                f32x4 v0, v1;
                asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(v0), "=&v"(v1)
                             : "v"(ep_r[pass]), "v"(ep_r[pass] ^ 16u)
                             : "memory");
                __builtin_amdgcn_sched_barrier(0);
                const bool ok = col_ok && m <= m_last;
                const int loc = ok ? (int)((m - m0) * ei.ldc) + loc_col : -1;
                const int64_t mc = m <= m_last ? m : m_last;
                epi_apply8<EPI>(ei, rs_c, rs_aux, loc, (const float*)ei.C + mc * ei.ldc + n_ld, v0, v1, b0, b1, ops[k % (PD + 1)], qmax);
                if (EK::may_colsum) {
                    const f32x4 z = {0, 0, 0, 0};
                    s0 += ok ? v0 : z;
                    s1 += ok ? v1 : z;
                    asm volatile("" : "+v"(s0), "+v"(s1));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (EK::colsum(ei)) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s0[j] += __shfl_xor(s0[j], 8, 64); s0[j] += __shfl_xor(s0[j], 16, 64); s0[j] += __shfl_xor(s0[j], 32, 64);
                s1[j] += __shfl_xor(s1[j], 8, 64); s1[j] += __shfl_xor(s1[j], 16, 64); s1[j] += __shfl_xor(s1[j], 32, 64);
            }
            if (lane_e < 8 && col_ok && m0 + wr * 64 < ei.M) {
                float* cp = ei.colpart + (2 * (int64_t)tm + wr) * ei.N + n;
                store4(cp, s0);
                store4(cp + 4, s1);
            }
        }
    }
}

template <bool BKM, int NTW, int EPI, int NLW = 4>
static void ws_launch_one(const bf16_t* a, int64_t lda, const bf16_t* b, int64_t ldb, int nk, int tiles_m, int tiles_n, int grid,
                          const EpiDev& e, hipStream_t s) {
    static bool attr_done = false;
    const int lds = WsCfg<NTW>::lds_bytes;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)gemm_ws_kernel<BKM, NTW, EPI, NLW>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_done = true;
    }
    gemm_ws_kernel<BKM, NTW, EPI, NLW><<<grid, 512 + 64 * NLW, lds, s>>>(a, lda, b, ldb, nk, tiles_m, tiles_n, e);
}
