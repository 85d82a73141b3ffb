// Persistent 256-row-tile bf16 MFMA GEMM for the large Linear launches of the training step
// (models/dit.py:118-155: qkv / proj / fc1 / fc2 forward, input gradient and weight gradient; 12 D^2 MAC per token and block).
//
// Why a second kernel next to gemm_bf16_kernel (gemm.hip): the 128 x 128 tile moves 32 KiB from L2 into LDS per 2.1 MFLOP;
// at two to three workgroups per CU that is 55-75 GB/s per CU, the ceiling of the vector-memory -> LDS path
// (MI355X_MICROARCH.md "Indexed rows: gather into LDS"), so its main loop stalls on LDS-DMA whatever the schedule.
// This kernel halves the staged bytes per MFMA and hides their latency completely:
//   * one 512-thread workgroup per CU, tile 256 x BN (BN = 256 or 192: 192 divides the 768 / 2304 / 3072 / 1152-wide
//     layers of DiT-B and DiT-XL into whole rounds of 256 workgroups), waves 2 (M) x 4 (N), wave tile 128 x BN/4;
//   * K in 64-deep tiles through TWO LDS stages; a stage is cut into 8 KiB parts (64 rows of A or B), each filled by one
//     LDS-DMA instruction per wave.  A K tile is computed in four phases (one 32-row quarter of each wave's A rows per
//     phase, the B fragments of the tile held in registers); every phase issues the DMA of two parts that lie 4-6 phases
//     ahead and waits with a COUNTED s_waitcnt vmcnt(N) only for the part the next phase reads, so 60-70 KiB stay in
//     flight per CU across raw s_barriers (cdna_hip_programming.md "Pipelining across barriers", 8-phase template);
//   * the two wave rows run half a phase apart (one extra barrier for waves 4-7 when a tile starts, one for waves 0-3 when
//     it ends): while one wave of a SIMD issues its MFMA cluster the other reads fragments and issues DMA;
//   * persistent: a workgroup walks a static list of (tile, K split) items; the DMA stream runs ahead across item
//     boundaries, so the first K tiles of the next item land during the epilogue of the current one;
//   * epilogue through a wave-private 4 KiB LDS image (no workgroup barrier): accumulators (held transposed: a lane owns 4
//     consecutive columns of one row) -> XOR-swizzled f32 rows -> 8 consecutive columns per lane, 16-byte global accesses,
//     the same epilogue arithmetic as gemm.hip (gemm_epi.h); column sums of the output (next bias gradient) stay in
//     registers and leave as one partial row per wave row (128 rows), folded by vaw_reduce_rows in a fixed order;
//   * split-K items write f32 slabs that splitk_reduce_kernel folds in a fixed order (deterministic, as in gemm.hip).
// Operand layouts: k-major ([rows][K], ds_read_b128) or mn-major ([K][rows], ds_read_b64_tr_b16), any combination.
//
// LDS images of one 8 KiB part:
//   k-major  [64 rows][128 B]   chunk' = chunk ^ ((row >> 1) & 7)                      (16-byte chunks, 8 per row)
//   mn-major [64 k][64 cols]    chunk' = chunk ^ ((((k >> 1) & 1) | (((k >> 3) & 1) << 1)) << 1)
// both conflict-free for the 16x16x32 fragment reads (bank = (addr/4) % 64; the swizzle is applied to the per-lane
// SOURCE address of the DMA and to the read address: cdna_hip_programming.md rule 21).
// A part i of a stage holds rows (or columns) {32 i .. 32 i + 31} and {128 + 32 i .. 128 + 32 i + 31} of the 256-row
// A tile: exactly what the two wave rows read in phase i.  B part p holds rows 64 p .. 64 p + 63.
#include <stdlib.h>

#include "gemm_epi.h"

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

#define P8_BM 256
#define P8_PART 8192
#define P8_EPI_BYTES 32768

template <int NTW> struct P8Cfg {
    static constexpr int BN = 64 * NTW;              // 4 waves x NTW MFMA tiles of 16 columns
    static constexpr int WN = 16 * NTW;
    static constexpr int a_bytes = 4 * P8_PART;
    static constexpr int stage_bytes = (4 + NTW) * P8_PART;
    static constexpr int lds_bytes = 2 * stage_bytes + P8_EPI_BYTES;
};

__device__ __forceinline__ int p8_mn_swz(int k) { return (((k >> 1) & 1) | (((k >> 3) & 1) << 1)) << 1; }

// Per-lane element offset (from the tile's first element at the current K position) of the 16 bytes this lane's
// LDS-DMA piece `wid` of part `p` fetches.  is_a: the A-part row set (two 32-row runs 128 apart), else 64 p + row.
template <bool KMAJOR>
__device__ __forceinline__ int p8_src_off(bool is_a, int p, int wid, int lane, int64_t ld, int valid) {
    const int r = 8 * wid + (lane >> 3);          // row (k-major) or k (mn-major) within the part
    if (KMAJOR) {
        const int chunk = (lane & 7) ^ ((r >> 1) & 7);
        int R = is_a ? (r < 32 ? 32 * p + r : 96 + 32 * p + r) : 64 * p + r;
        R = R < valid ? R : valid - 1;            // LDS-DMA cannot zero-fill: rows beyond the edge mirror a valid row
        return (int)(R * ld) + chunk * 8;
    } else {
        const int chunk = (lane & 7) ^ p8_mn_swz(r);
        int col = is_a ? (chunk < 4 ? 32 * p + 8 * chunk : 96 + 32 * p + 8 * chunk) : 64 * p + 8 * chunk;
        col = col < valid ? col : 0;
        return (int)(r * ld) + col;
    }
}

// 16 (rows r16 .. r16+15 of the part) x 32 (k sub-step s) operand fragment.
template <bool KMAJOR>
__device__ __forceinline__ bf16x8 p8_frag(const char* part, int r16, int s, int lane) {
    if (KMAJOR) {
        const int row = r16 + (lane & 15);
        const int chunk = (4 * s + (lane >> 4)) ^ ((row >> 1) & 7);
        return *reinterpret_cast<const bf16x8*>(part + row * 128 + (chunk << 4));
    } else {
        const int li = lane & 15, q = li >> 2, p = li & 3;
        const int kb = 32 * s + 8 * (lane >> 4) + q;
        const int ch = ((r16 >> 3) + (p >> 1)) ^ p8_mn_swz(kb);
        const char* a0 = part + kb * 128 + (ch << 4) + 8 * (p & 1);
        bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)a0);
        bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(a0 + 512));   // k + 4: same swizzle
        bf16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return r;
    }
}

template <bool AK, bool BKM, int NTW>
__global__ void __launch_bounds__(512, 2)
gemm_p8_kernel(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ B, int64_t ldb, int nk_total,
               int tiles_m, int tiles_n, int n_split, EpiDev e) {
    using Cfg = P8Cfg<NTW>;
    constexpr int LS = 4 + NTW;                                  // DMA pieces per wave and K tile
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 stages][A parts 0-3 | B parts] | 8 x 4 KiB epilogue images
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wid >> 2, wc = wid & 3;
    // ---- static item list: G resident workgroups; XCD x (workgroups b = x mod 8) owns a contiguous run of every round ----
    const int G = gridDim.x, n_tiles = tiles_m * tiles_n, n_items = n_tiles * n_split;
    int it_cur;
    {
        const int b = blockIdx.x, x = b & 7, q = G >> 3, r = G & 7;
        it_cur = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
    }
    const int nk_per = (nk_total + n_split - 1) / n_split;
    const int64_t a_step = AK ? 64 : 64 * lda, b_step = BKM ? 64 : 64 * ldb;

    // ---- the DMA stream (runs ahead of the MFMAs; its own item / K-tile position) ----
    int iss_item = it_cur, iss_kt = 0, iss_nk = 0, iss_stage = 0;
    bool iss_done = iss_item >= n_items;
    const bf16_t *iss_a = A, *iss_b = B;
    int off_a[4], off_b[NTW];
    auto iss_open = [&]() {                       // position the stream on the first K tile of item iss_item
        const int split = iss_item / n_tiles, tile = iss_item - split * n_tiles;
        const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
        const int64_t m0 = (int64_t)tm * P8_BM, n0 = (int64_t)tn * Cfg::BN;
        const int kt0 = split * nk_per;
        iss_nk = kt0 + nk_per <= nk_total ? nk_per : nk_total - kt0;
        iss_kt = 0;
        const int mvalid = e.M - m0 < P8_BM ? (int)(e.M - m0) : P8_BM;
        const int nvalid = e.N - n0 < Cfg::BN ? (int)(e.N - n0) : Cfg::BN;
        iss_a = (AK ? A + m0 * lda : A + m0) + kt0 * a_step;
        iss_b = (BKM ? B + n0 * ldb : B + n0) + kt0 * b_step;
#pragma unroll
        for (int p = 0; p < 4; ++p) off_a[p] = p8_src_off<AK>(true, p, wid, lane, lda, mvalid);
#pragma unroll
        for (int p = 0; p < NTW; ++p) off_b[p] = p8_src_off<BKM>(false, p, wid, lane, ldb, nvalid);
    };
    if (!iss_done) iss_open();
    // piece c of the stream order [B parts 0..NTW-1, A parts 0..3]
    auto iss_piece = [&](auto cc) {
        constexpr int c = decltype(cc)::value;
        if (iss_done) return;
        char* dst = smem + iss_stage * Cfg::stage_bytes + wid * 1024;
        if (c < NTW) __builtin_amdgcn_global_load_lds((gbl_ptr_t)(iss_b + off_b[c < NTW ? c : 0]), (lds_ptr_t)(dst + Cfg::a_bytes + c * P8_PART), 16, 0, 0);
        else __builtin_amdgcn_global_load_lds((gbl_ptr_t)(iss_a + off_a[c >= NTW ? c - NTW : 0]), (lds_ptr_t)(dst + (c - NTW) * P8_PART), 16, 0, 0);
    };
    auto iss_advance = [&]() {                    // after the last piece of a K tile
        if (iss_done) return;
        iss_stage ^= 1;
        iss_a += a_step;
        iss_b += b_step;
        if (++iss_kt == iss_nk) {
            iss_item += G;
            if (iss_item >= n_items) iss_done = true;
            else iss_open();
        }
    };
#define P8_PIECE(c) iss_piece(std::integral_constant<int, (c)>{})
    // issue slots of the four phases (stream positions; NTW = 4: 2,2,2,2; NTW = 3: 2,2,2,1)
    auto issue_ph1 = [&]() { P8_PIECE(4); P8_PIECE(5); };
    auto issue_ph2 = [&]() { P8_PIECE(6); if (LS == 8) P8_PIECE(LS - 1); iss_advance(); };
    auto issue_ph3 = [&]() { P8_PIECE(0); P8_PIECE(1); };
    auto issue_ph4 = [&]() { P8_PIECE(2); P8_PIECE(3); };
    // counted waits: pieces younger than the one the NEXT phase reads (derivation in DESIGN.md §5)
#define P8_WAIT(n_full)                                                                     \
    do {                                                                                    \
        if (iss_done) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                      \
        else asm volatile("s_waitcnt vmcnt(" #n_full ")" ::: "memory");                     \
    } while (0)

    // prologue: K tile 0 of the stream completely, the ph3/ph4 slots of K tile 1
    if (!iss_done) {
        P8_PIECE(0); P8_PIECE(1); P8_PIECE(2); P8_PIECE(3); P8_PIECE(4); P8_PIECE(5); P8_PIECE(6);
        if (LS == 8) P8_PIECE(LS - 1);
        iss_advance();
        issue_ph3();
        issue_ph4();
    }
    P8_WAIT(7);
    __builtin_amdgcn_s_barrier();

    int stage = 0;
    const int wn0 = wc * Cfg::WN;
    for (; it_cur < n_items; it_cur += G) {
        const int split = it_cur / n_tiles, tile = it_cur - split * n_tiles;
        const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
        const int64_t m0 = (int64_t)tm * P8_BM, n0 = (int64_t)tn * Cfg::BN;
        const int kt0 = split * nk_per;
        const int nk = kt0 + nk_per <= nk_total ? nk_per : nk_total - kt0;
        f32x4 acc[8][NTW];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int u = 0; u < NTW; ++u) acc[i][u] = f32x4{0, 0, 0, 0};
        if (wr == 1) __builtin_amdgcn_s_barrier();       // waves 4-7 run half a phase behind waves 0-3
        for (int kt = 0; kt < nk; ++kt) {
            const char* st = smem + stage * Cfg::stage_bytes;
            stage ^= 1;
            bf16x8 bfr[2][NTW], af[2][2];
            auto load_a = [&](int j) {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int s = 0; s < 2; ++s) af[s][t] = p8_frag<AK>(st + j * P8_PART, wr * 32 + 16 * t, s, lane);
            };
#define P8_MMA(j)                                                                                                     \
    do {                                                                                                              \
        __builtin_amdgcn_s_setprio(1);                                                                                \
        _Pragma("unroll") for (int s = 0; s < 2; ++s)                                                                 \
            _Pragma("unroll") for (int t = 0; t < 2; ++t)                                                             \
                _Pragma("unroll") for (int u = 0; u < NTW; ++u)                                                       \
                    acc[2 * (j) + t][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[s][u], af[s][t], acc[2 * (j) + t][u], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);                                                                                \
    } while (0)
            // ---- phase 1: B fragments of the whole K tile + A quarter 0
#pragma unroll
            for (int u = 0; u < NTW; ++u) {
                const int n = wn0 + 16 * u;
#pragma unroll
                for (int s = 0; s < 2; ++s) bfr[s][u] = p8_frag<BKM>(st + Cfg::a_bytes + (n >> 6) * P8_PART, n & 63, s, lane);
            }
            load_a(0);
            issue_ph1();
            P8_WAIT(8);
            __builtin_amdgcn_s_barrier();
            P8_MMA(0);
            __builtin_amdgcn_s_barrier();
            // ---- phase 2
            load_a(1);
            issue_ph2();
            if (LS == 8) P8_WAIT(9); else P8_WAIT(8);
            __builtin_amdgcn_s_barrier();
            P8_MMA(1);
            __builtin_amdgcn_s_barrier();
            // ---- phase 3
            load_a(2);
            issue_ph3();
            if (LS == 8) P8_WAIT(10); else P8_WAIT(9);
            __builtin_amdgcn_s_barrier();
            P8_MMA(2);
            __builtin_amdgcn_s_barrier();
            // ---- phase 4
            load_a(3);
            issue_ph4();
            P8_WAIT(7);
            __builtin_amdgcn_s_barrier();
            P8_MMA(3);
            __builtin_amdgcn_s_barrier();
        }
        if (wr == 0) __builtin_amdgcn_s_barrier();       // level the two wave rows: both run their epilogues together

        if (e.debug == 1) {   // measurement only (VAW_GEMM_DEBUG=1): no epilogue; one never-taken store keeps the accumulators alive
            float t = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int u = 0; u < NTW; ++u) t += acc[i][u][0] + acc[i][u][1] + acc[i][u][2] + acc[i][u][3];
            if (t == 12345.678f) ((float*)e.C)[0] = t;
            continue;
        }
        // ---- epilogue: 8 row tiles of 16 rows through this wave's private LDS image ----
        char* ep = smem + 2 * Cfg::stage_bytes + wid * 4096;
        const int wr_row = lane & 15, wr_g = lane >> 4;                 // accumulator layout: row, group of 4 columns
        const int rd_row = lane >> 3, rd_c8 = lane & 7;                 // read-back layout: row within 8, group of 8 columns
        const int64_t n = n0 + wn0 + 8 * rd_c8;
        const bool col_ok = 8 * rd_c8 < Cfg::WN && n < e.N;             // N % 8 == 0: a group is in or out as a whole
        f32x4 b0 = {0, 0, 0, 0}, b1 = {0, 0, 0, 0};
        if (e.bias && n_split == 1 && col_ok) {
            b0 = load4(e.bias + n);
            b1 = load4(e.bias + n + 4);
        }
        f32x4 s0 = {0, 0, 0, 0}, s1 = {0, 0, 0, 0};
        float* slab = n_split > 1 ? e.slab + (int64_t)split * e.M * e.N : nullptr;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
            for (int u = 0; u < NTW; ++u)
                *reinterpret_cast<f32x4*>(ep + wr_row * 256 + (((4 * u + wr_g) ^ wr_row) << 4)) = acc[i][u];
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                const int rr = pass * 8 + rd_row;
                f32x4 v0 = *reinterpret_cast<const f32x4*>(ep + rr * 256 + (((2 * rd_c8) ^ rr) << 4));
                f32x4 v1 = *reinterpret_cast<const f32x4*>(ep + rr * 256 + (((2 * rd_c8 + 1) ^ rr) << 4));
                const int64_t m = m0 + wr * 128 + 16 * i + rr;
                if (!col_ok || m >= e.M) continue;
                if (n_split > 1) {
                    float* dst = slab + m * e.N + n;
                    store4(dst, v0);             // split-K partials are re-read at once by the reduce: keep them cached
                    store4(dst + 4, v1);
                } else {
                    epi_row8(e, (unsigned)m, n, v0, v1, b0, b1);
                    s0 += v0;
                    s1 += v1;
                }
            }
        }
        if (e.colpart && n_split == 1) {
            // 8 row groups of the read-back layout: fold lane bits 3, 4, 5 in a fixed order; one partial row per wave row
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s0[j] += __shfl_xor(s0[j], 8, 64); s0[j] += __shfl_xor(s0[j], 16, 64); s0[j] += __shfl_xor(s0[j], 32, 64);
                s1[j] += __shfl_xor(s1[j], 8, 64); s1[j] += __shfl_xor(s1[j], 16, 64); s1[j] += __shfl_xor(s1[j], 32, 64);
            }
            if (lane < 8 && col_ok && m0 + wr * 128 < e.M) {
                float* cp = e.colpart + (2 * (int64_t)tm + wr) * e.N + n;
                store4(cp, s0);
                store4(cp + 4, s1);
            }
        }
    }
}

// ---- host side ----------------------------------------------------------------------------------------------
// Tile width and split count for a shape, or use = false when the 128 x 128 kernel of gemm.hip should keep it.
struct P8Plan {
    bool use;
    int ntw, split, grid;
};

static int p8_num_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

// efficiency of running `items` equal work items on `cus` workgroups, times the share of the tile that is real output
static double p8_fill(int64_t M, int64_t N, int bn, int split, int cus) {
    const int64_t tm = (M + 255) / 256, tn = (N + bn - 1) / bn, items = tm * tn * split;
    const int64_t rounds = (items + cus - 1) / cus;
    return (double)items / (double)(rounds * cus) * ((double)M * N / ((double)tm * 256 * tn * bn));
}

P8Plan vaw_p8_plan(int64_t M, int64_t N, int64_t K, bool plain_f32, bool want_colsum, int64_t ws_floats, int force) {
    P8Plan pl{false, 4, 1, 0};
    if (force == 0) return pl;
    const int cus = p8_num_cus();
    const int nk = (int)(K / 64);
    double best = 0.0;
    for (int ntw = 4; ntw >= 3; --ntw) {
        if (force == 2 && ntw != 4) continue;
        if (force == 3 && ntw != 3) continue;
        const int bn = 64 * ntw;
        const int64_t tiles = ((M + 255) / 256) * ((N + bn - 1) / bn);
        int smax = 1;
        if (plain_f32 && !want_colsum && ws_floats > 0) {
            int64_t s = nk / 4;                                  // every split keeps >= 4 K tiles (256 of K)
            if (s > ws_floats / (M * N)) s = ws_floats / (M * N);
            if (s > 64) s = 64;
            if (s > 2 * cus / tiles) s = 2 * cus / tiles;        // no point beyond two rounds
            smax = s < 1 ? 1 : (int)s;
        }
        for (int s = 1; s <= smax; ++s) {
            const int per = (nk + s - 1) / s;
            if ((nk + per - 1) / per != s) continue;             // no empty splits
            // slab traffic costs: one f32 tile written and re-read per item vs 2*per K tiles of MFMA work
            double f = p8_fill(M, N, bn, s, cus);
            if (s > 1) f *= (double)per / (per + 3.0);
            if (f > best + 1e-9) { best = f; pl.ntw = ntw; pl.split = s; }
        }
    }
    const int bn = 64 * pl.ntw;
    const int64_t items = ((M + 255) / 256) * ((N + bn - 1) / bn) * pl.split;
    pl.grid = (int)(items < cus ? items : cus);
    pl.use = force > 0 || (best >= 0.70 && items >= cus / 2 && nk >= 4);
    return pl;
}

template <bool AK, bool BKM, int NTW>
static void p8_launch_one(const bf16_t* a, int64_t lda, const bf16_t* b, int64_t ldb, int nk, int tiles_m, int tiles_n, int split,
                          int grid, const EpiDev& e, hipStream_t s) {
    static bool attr_done = false;
    const int lds = P8Cfg<NTW>::lds_bytes;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)gemm_p8_kernel<AK, BKM, NTW>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_done = true;
    }
    gemm_p8_kernel<AK, BKM, NTW><<<grid, 512, lds, s>>>(a, lda, b, ldb, nk, tiles_m, tiles_n, split, e);
}

void vaw_p8_launch(const P8Plan& pl, int a_kmajor, int b_kmajor, int64_t M, int64_t N, int64_t K, const bf16_t* a, int64_t lda,
                   const bf16_t* b, int64_t ldb, const EpiDev& e, hipStream_t s) {
    const int bn = 64 * pl.ntw;
    const int tiles_m = (int)((M + 255) / 256), tiles_n = (int)((N + bn - 1) / bn), nk = (int)(K / 64);
#define P8_GO(AKv, BKv)                                                                                       \
    do {                                                                                                      \
        if (pl.ntw == 4) p8_launch_one<AKv, BKv, 4>(a, lda, b, ldb, nk, tiles_m, tiles_n, pl.split, pl.grid, e, s); \
        else p8_launch_one<AKv, BKv, 3>(a, lda, b, ldb, nk, tiles_m, tiles_n, pl.split, pl.grid, e, s);       \
    } while (0)
    if (a_kmajor && b_kmajor) P8_GO(true, true);
    else if (a_kmajor && !b_kmajor) P8_GO(true, false);
    else if (!a_kmajor && b_kmajor) P8_GO(false, true);
    else P8_GO(false, false);
}
