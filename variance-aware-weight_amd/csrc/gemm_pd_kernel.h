// Persistent bf16 MFMA GEMM whose epilogue DRAINS UNDER THE NEXT TILE'S K LOOP ("parked drain"), for the forward and input-gradient
// launches of the Linear layers (models/dit.py:118-155: qkv / proj / fc1 / fc2; K = 768 .. 3072, M = B*T rows).
//
// Why a third kernel next to gemm_p8_kernel: there a workgroup owns its CU, so while it runs the epilogue of a 256 x 256 tile the
// MFMA pipes idle, and with all 256 CUs storing at once that epilogue is HBM-write-bound (fc1 forward: 10.8 us of every 32 us item).
// The epilogue cannot simply be issued and left to drain: vmcnt retires in order and counts stores, so the K loop's counted waits for
// its LDS-DMA pieces would wait for every store in front of them.  This kernel makes the order work for it instead:
//   * tile 128 x BN (BN = 256 | 192), 8 waves as 2 (M) x 4 (N), wave tile 64 x BN/4 = 64 accumulator registers: room to PARK the
//     finished tile in registers as packed bf16 (32 registers: the value acc*alpha + bias rounded to bf16 -- what the reference's
//     autocast Linear hands on -- is all the epilogue kinds of the training step need) and to start the next tile's K loop at once;
//   * the parked tile leaves in STEPS drain slots, one per K tile of the NEXT item: a slot stages 16 rows through a wave-private
//     2 KiB LDS image (accumulator layout -> row layout, 16-byte global accesses), applies the epilogue arithmetic (gemm_epi.h kinds
//     STORE / GELU / DGELU / GATE) and issues its stores; operand loads (GELU' argument, residual, gate) are issued two slots ahead
//     by inline asm into registers that stay allocated;
//   * every drain slot issues the SAME number of vector-memory operations (padding with out-of-range stores, which the buffer
//     hardware drops), so the K loop's counted waits stay exact: N = pieces younger than the one needed + ND per drain slot among them
//     (selected by a scalar branch on how many of the last slots were live);
//   * the LDS that the 256-row kernel spends on epilogue images goes to the ring instead: THREE stages of (2 + NTW) x 8 KiB, every
//     piece issued two K tiles ahead of its use, 7-9 KiB per wave in flight across the raw barriers;
//   * two phases per K tile (A rows 0-31 / 32-63 of the wave's 64; 4 NTW MFMAs each), wave rows staggered by half a phase as in
//     gemm_p8_kernel, same LDS images, fragment reads and swizzles (p8_frag, p8_src_off with the 128-row part map).
// Bytes staged per MFMA are 1.5 x the 256-row kernel's; what it buys is an epilogue that costs issue slots instead of idle MFMA time.
// Numerics: identical to gemm_p8_kernel for STORE / GELU / GATE (they round acc*alpha + bias to bf16 first anyway); the GELU' kind
// multiplies the bf16-rounded input gradient by GELU'(h) -- the two roundings of the reference's autocast (matmul output bf16, then
// the GELU backward), where gemm_p8_kernel rounds once.  Column sums are taken from the stored values in a fixed order (one partial row per
// 64 rows).
#pragma once
#include "gemm_p8_kernel.h"

#define PD_BM 128
// measurement / bisecting builds only (make exp XSRC="gemm_pd gemm_pd_dgrad" XF=-DPD_DBG=n): 1 = K loops only, nothing parked or stored;
// 2 = no bias load; 3 = parked tiles leave in the open (no drain slots inside the K loop); 4 = every drain store out of range (dropped)
#ifndef PD_DBG
#define PD_DBG 0
#endif
// 1: ONE barrier per K tile.  With three ring stages a piece is issued into the stage that was read a whole K tile ago, so the
// barrier that makes a K tile's pieces visible to every wave also proves that everybody has left the stage the next pieces go
// into: wait -> barrier -> issue the whole set two K tiles ahead -> all fragment reads of the K tile -> 8 NTW MFMAs, the waves
// free-running in between (no phase structure, no wave-row stagger).  0: the two-phase, four-barrier form.
#ifndef PD_ONEBAR
#define PD_ONEBAR 0
#endif
template <int NTW> struct PdCfg {
    static constexpr int BN = 64 * NTW, WN = 16 * NTW;
    static constexpr int LS = 2 + NTW;                      // DMA pieces per wave and K tile: B parts 0..NTW-1, A parts 0, 1
    static constexpr int a_bytes = 2 * P8_PART;
    static constexpr int stage_bytes = LS * P8_PART;
    static constexpr int ring_bytes = 3 * stage_bytes;
    static constexpr int lds_bytes = ring_bytes + 8 * 2048;  // + one 16-row x 128-byte staging image per wave
};

// Operand loads the compiler knows nothing about, into FIXED physical registers v228 .. v255 that the kernel keeps out of the
// register allocator's hands (amdgpu_num_vgpr(228) on the kernel; one clobber of v255 makes the descriptor count them).  A value the
// compiler manages would travel through PHI copies at the drain switch and the loop headers -- v_mov of a register whose load is
// still in flight reads garbage (seen in the first build's ISA); a fixed register is only ever touched by the load and by the
// v_mov that fetches it after the counted wait that covers it.  Uniform 64-bit base in scalar registers + 32-bit per-lane byte
// offset; lanes with nothing to load pass offset 0 (the base itself is always a valid address).
#define PD_VGPR_CAP 228
#define PD_R_BIAS 228
#define PD_R_X0 232      /* x = GELU' argument | gate: two slots (step parity) of 4 registers */
#define PD_R_Y0 240      /* y = residual: two slots of 2 x 4 registers */
__device__ __forceinline__ const char* pd_uniform(const void* p) {       // the address as two scalar words
    const uint64_t b = (uint64_t)(uintptr_t)p;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32));
    return (const char*)(uintptr_t)(((uint64_t)hi << 32) | lo);
}
template <int R> __device__ __forceinline__ void pd_load16(const void* base, unsigned voff) {
    static_assert(R >= PD_VGPR_CAP && R + 3 <= 255, "reserved registers");
    // (s_nop 4: a VALU write of the base SGPRs -- v_readlane of a spilled pointer -- needs 5 wait states before a VMEM instruction
    //  reads them as its address; the compiler pads its own instructions, it cannot see into inline asm)
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 v[%2:%3], %0, %1" ::"v"(voff), "s"(base), "n"(R), "n"(R + 3) : "memory");
}
template <int R> __device__ __forceinline__ void pd_load4(const void* base, unsigned voff) {
    static_assert(R >= PD_VGPR_CAP && R <= 255, "reserved registers");
    asm volatile("s_nop 4\n\tglobal_load_dword v[%2], %0, %1" ::"v"(voff), "s"(base), "n"(R) : "memory");
}
template <int R> __device__ __forceinline__ f32x4 pd_take16() {          // after the counted wait that covers the load
    f32x4 v;
    asm volatile("v_mov_b32 %0, v[%4]\n\tv_mov_b32 %1, v[%5]\n\tv_mov_b32 %2, v[%6]\n\tv_mov_b32 %3, v[%7]"
                 : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3])
                 : "n"(R), "n"(R + 1), "n"(R + 2), "n"(R + 3));
    return v;
}
template <int R> __device__ __forceinline__ float pd_take4() {
    float v;
    asm volatile("v_mov_b32 %0, v[%1]" : "=v"(v) : "n"(R));
    return v;
}
// a vector-memory operation that only counts: a store at an out-of-range buffer offset, which the hardware drops.  Inline asm --
// the optimizer removes all but the last of several identical builtin stores (dead-store elimination), and the slot would then
// issue fewer operations than the counted waits assume (found as stale operands in the first GATE / GELU' runs).
__device__ __forceinline__ void pd_dummy_op(__amdgpu_buffer_rsrc_t rs) {
    const unsigned oob = EPI_OOB;
    const float z = 0.f;
    asm volatile("s_nop 4\n\tbuffer_store_dword %0, %1, %2, 0 offen" ::"v"(z), "v"(oob), "s"(rs) : "memory");
}
template <int N> __device__ __forceinline__ void pd_vmwait() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// wait until all but the BASE + c * ND youngest vector-memory operations have completed (c = live drain slots among them: 0, 1, 2)
template <int BASE, int ND> __device__ __forceinline__ void pd_wait(int c) {
    if (ND == 0 || c == 0) pd_vmwait<BASE>();
    else if (c == 1) pd_vmwait<BASE + ND>();
    else pd_vmwait<BASE + 2 * ND>();
}
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pd_pack2(float a, float b) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, bf16x2_t{(bf16_t)a, (bf16_t)b});
}
__device__ __forceinline__ float pd_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float pd_hi(unsigned w) { return __uint_as_float(w & 0xFFFF0000u); }

// per-lane source offset of A part p (rows {32 p .. 32 p + 31} and {64 + 32 p ..} of the 128-row tile), k-major A only
__device__ __forceinline__ unsigned pd_src_off_a(int p, int wid, int lane, int64_t ld, int valid) {
    const int r = 8 * wid + (lane >> 3);
    const int chunk = (lane & 7) ^ ((r >> 1) & 7);
    int R = r < 32 ? 32 * p + r : 32 + 32 * p + r;
    R = R < valid ? R : valid - 1;
    return (unsigned)(R * ld * 2) + chunk * 16;
}

template <typename F, int... Js> __device__ __forceinline__ void pd_unroll(F&& f, std::integer_sequence<int, Js...>) {
    (f(std::integral_constant<int, Js>{}), ...);
}

template <bool BKM, int NTW, int EPI>
__global__ void __launch_bounds__(512, 2) __attribute__((amdgpu_num_vgpr(PD_VGPR_CAP)))
gemm_pd_kernel(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ B, int64_t ldb, int nk, int tiles_m, int tiles_n,
               EpiDev e) {
    static_assert(EPI == P8_STORE || EPI == P8_GELU || EPI == P8_DGELU || EPI == P8_GATE, "epilogue kinds of the Linear launches");
    using Cfg = PdCfg<NTW>;
    using EK = EpiKind<EPI>;
    constexpr int LS = Cfg::LS, WN = Cfg::WN;
    // drain geometry: STEPS slots of 64 / STEPS rows per wave tile; NL operand loads and NS stores per slot; operands LEAD slots ahead
    constexpr int STEPS = EPI == P8_STORE ? 4 : 8;
    constexpr int NL = EPI == P8_GATE ? 3 : EPI == P8_DGELU ? 1 : 0;
    constexpr int NS = EPI == P8_GATE ? 4 : EPI == P8_GELU ? 2 : EPI == P8_DGELU ? 1 : 2;
    constexpr int ND = NL + NS;
    constexpr int LEAD = NL > 0 ? 2 : 0;
    constexpr int DMAX = STEPS + LEAD;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [3 stages][A parts 0-1 | B parts] | 8 x 2 KiB staging images
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wid >> 2, wc = wid & 3;
    const int G = gridDim.x, n_items = tiles_m * tiles_n;
    int it_cur;
    {
        const int b = blockIdx.x, x = b & 7, q = G >> 3, r = G & 7;
        it_cur = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
    }
    // item -> (row tile, column tile): wide outputs in groups of 8 row tiles with the column tile slow inside a group, so the 32
    // consecutive items an XCD works on in a round form an 8 x 4 block of tiles (8 A panels of 128 rows + 4 B panels through its L2)
    auto decode = [&](int it, int& tm, int& tn) __attribute__((always_inline)) {
        if (tiles_n >= 8 && e.debug != 3) {
            const int gsz = 8 * tiles_n, gi = it / gsz, within = it - gi * gsz;
            const int rows = tiles_m - 8 * gi < 8 ? tiles_m - 8 * gi : 8;
            tn = within / rows;
            tm = 8 * gi + within - tn * rows;
        } else {
            tm = it / tiles_n;
            tn = it - tm * tiles_n;
        }
    };

    // ---- the DMA stream: its own item / K position, two K tiles ahead of the MFMAs.  Once it has run out of items it keeps
    // issuing pieces with out-of-range offsets (the hardware writes zeros into a stage nobody reads), so the counted waits of
    // the K loop need no second form ----
    int iss_item = it_cur, iss_kt = 0, iss_stage = 0;
    __amdgpu_buffer_rsrc_t rs_a = epi_rsrc(A), rs_b = epi_rsrc(B);
    unsigned so_a = 0, so_b = 0;
    const unsigned step_a = 128u, step_b = BKM ? 128u : (unsigned)(64 * ldb * 2);
    unsigned off_a[2], off_b[NTW];
    auto iss_open = [&]() __attribute__((always_inline)) {
        if (iss_item >= n_items) {
#pragma unroll
            for (int p = 0; p < 2; ++p) off_a[p] = EPI_OOB;
#pragma unroll
            for (int p = 0; p < NTW; ++p) off_b[p] = EPI_OOB;
            so_a = so_b = 0;
            return;
        }
        int tm, tn;
        decode(iss_item, tm, tn);
        const int64_t m0 = (int64_t)tm * PD_BM, n0 = (int64_t)tn * Cfg::BN;
        const int mvalid = e.M - m0 < PD_BM ? (int)(e.M - m0) : PD_BM;
        const int nvalid = e.N - n0 < Cfg::BN ? (int)(e.N - n0) : Cfg::BN;
        iss_kt = 0;
        so_a = so_b = 0;
        rs_a = epi_rsrc(A + m0 * lda);
        rs_b = epi_rsrc(BKM ? B + n0 * ldb : B + n0);
#pragma unroll
        for (int p = 0; p < 2; ++p) off_a[p] = pd_src_off_a(p, wid, lane, lda, mvalid);
#pragma unroll
        for (int p = 0; p < NTW; ++p) off_b[p] = p8_src_off<BKM>(false, p, wid, lane, ldb, nvalid);
    };
    iss_open();
    auto iss_piece = [&](auto cc) __attribute__((always_inline)) {          // piece c of the stream order [B parts 0 .. NTW-1, A parts 0, 1]
        constexpr int c = decltype(cc)::value;
        char* dst = smem + iss_stage * Cfg::stage_bytes + wid * 1024;
        if (c < NTW) p8_dma16(rs_b, dst + Cfg::a_bytes + c * P8_PART, off_b[c < NTW ? c : 0], so_b);
        else p8_dma16(rs_a, dst + (c >= NTW ? c - NTW : 0) * P8_PART, off_a[c >= NTW ? c - NTW : 0], so_a);
    };
    auto iss_advance = [&]() __attribute__((always_inline)) {
        iss_stage = iss_stage == 2 ? 0 : iss_stage + 1;
        so_a += step_a;
        so_b += step_b;
        if (++iss_kt == nk) {
            iss_item += G;
            iss_open();
        }
    };
#define PD_PIECE(c) iss_piece(std::integral_constant<int, (c)>{})
    auto issue_p1 = [&]() __attribute__((always_inline)) { PD_PIECE(0); PD_PIECE(1); PD_PIECE(2); };
    auto issue_p2 = [&]() __attribute__((always_inline)) { PD_PIECE(3); PD_PIECE(4); if (LS == 6) PD_PIECE(LS - 1); iss_advance(); };
    // prologue: sets 0 and 1 (the host guarantees nk >= 4: both belong to the first item)
    issue_p1(); issue_p2();
    issue_p1(); issue_p2();
    pd_vmwait<1 + LS>();                   // B parts and A part 0 of set 0 (younger: its A part 1 and set 1)
    __builtin_amdgcn_s_barrier();

    // ---- the parked tile and its drain state ----
    unsigned pk[4][NTW][2];                // [row tile of 16][column tile of 16] -> 4 bf16 of this lane (row l & 15, columns 4 (l >> 4) ..)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int u = 0; u < NTW; ++u) pk[i][u][0] = pk[i][u][1] = 0u;
    bool parked = false;                   // a finished tile waits in pk[] (every item of this workgroup but the first)
    int64_t pm0 = 0, pn0 = 0;              // origin of the parked tile
    int ptm = 0;
    __amdgpu_buffer_rsrc_t rs_c = epi_rsrc(e.C), rs_aux = epi_rsrc(e.C);
    const char *rl_x = pd_uniform(e.C), *rl_y = pd_uniform(e.C);     // operand loads: x = GELU' argument | gate, y = residual
    unsigned g_smp = 0, g_rin = 0;         // GATE: (sample, row in sample) of the next step whose operands are loaded
    asm volatile("" ::: "v255");           // the reserved registers v228 .. v255 belong to this kernel's allocation
    f32x4 cs0 = {0, 0, 0, 0}, cs1 = {0, 0, 0, 0};                          // column sums of the parked tile (STORE / DGELU)
    char* const stg = smem + Cfg::ring_bytes + wid * 2048;
    const unsigned stg_base = (unsigned)(uintptr_t)(lds_ptr_t)stg;
    const f32x4 zero4 = {0, 0, 0, 0};

    // one drain slot.  Slot D: operand loads of step D (D < STEPS), then step D - LEAD.  Always NL + NS vector-memory operations.
    auto run_slot = [&](auto dd, auto ff) __attribute__((always_inline)) {
        constexpr int D = decltype(dd)::value;
        constexpr bool flush = decltype(ff)::value;
        constexpr bool do_ld = NL > 0 && D < STEPS;
        constexpr bool do_ex = D >= LEAD && D < DMAX;
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        const int mrem = (int)(e.M - pm0) - wr * 64;          // valid rows of this wave's 64
        // the operands of the step this slot runs were loaded two slots ago, into the registers that THIS slot's loads (same step
        // parity) are about to overwrite: wait for them and fetch them first.  Younger than those loads: the stores of that
        // slot, the slot in between and, inside the K loop, the pieces of two K tiles (in flush mode the slots run back to back)
        f32x4 op_x = {0, 0, 0, 0}, op_y0 = {0, 0, 0, 0}, op_y1 = {0, 0, 0, 0};
        if (NL > 0 && do_ex) {
            constexpr int par = (D - LEAD) & 1;
            if (flush) pd_vmwait<NS + ND>();
            else pd_vmwait<2 * LS + NS + ND>();
            op_x = pd_take16<PD_R_X0 + 4 * par>();
            if (EPI == P8_GATE) {
                op_y0 = pd_take16<PD_R_Y0 + 8 * par>();
                op_y1 = pd_take16<PD_R_Y0 + 8 * par + 4>();
            }
        }
        if (NL > 0) {
            if (do_ld) {
                constexpr int S = D < STEPS ? D : 0, par = S & 1;
                if (EPI == P8_DGELU) {
                    // GELU' argument of rows 8 S + (lane >> 3), columns 8 (lane & 7) ..: 16 bytes
                    const int row = 8 * S + (lane_e >> 3), c8 = lane_e & 7;
                    const bool ok = 8 * c8 < WN && pn0 + wc * WN + 8 * c8 < e.N && row < mrem;
                    pd_load16<PD_R_X0 + 4 * par>(rl_x, ok ? 2u * (unsigned)((wr * 64 + row) * e.ldc + wc * WN + 8 * c8) : 0u);
                } else {   // P8_GATE: gate of the step's sample (rows_per_batch % 8 == 0: one sample per step), residual of 2 x 4 rows
                    const int c4 = lane_e & 15;
                    const bool cok = 4 * c4 < WN && pn0 + wc * WN + 4 * c4 < e.N;
                    pd_load16<PD_R_X0 + 4 * par>(rl_x, (cok && 8 * S < mrem) ? 4u * (unsigned)(g_smp * e.gate_ld + pn0 + wc * WN + 4 * c4) : 0u);
                    {
                        const int row0 = 8 * S + (lane_e >> 4), row1 = row0 + 4;
                        pd_load16<PD_R_Y0 + 8 * par>(rl_y, (cok && row0 < mrem) ? 4u * (unsigned)((wr * 64 + row0) * e.ldc + wc * WN + 4 * c4) : 0u);
                        pd_load16<PD_R_Y0 + 8 * par + 4>(rl_y, (cok && row1 < mrem) ? 4u * (unsigned)((wr * 64 + row1) * e.ldc + wc * WN + 4 * c4) : 0u);
                    }
                    g_rin += 8;
                    if (g_rin >= (unsigned)e.rpb) { g_rin -= (unsigned)e.rpb; g_smp += 1; }
                }
            } else {
#pragma unroll
                for (int j = 0; j < NL; ++j) pd_dummy_op(rs_c);
            }
        }
        if (!do_ex) {
#pragma unroll
            for (int j = 0; j < NS; ++j) pd_dummy_op(rs_c);
            return;
        }
        constexpr int S = do_ex ? D - LEAD : 0;               // the step
        constexpr int RPS = 64 / STEPS;                       // rows per step: 16 or 8
        constexpr int I = S * RPS / 16;                       // row tile of 16 it belongs to
        constexpr int HH = RPS == 8 ? (S & 1) : 0;            // which half of the staged row tile
        if (RPS == 16 || HH == 0) {
            // stage row tile I: lane (r = l & 15, g = l >> 4) writes its 4 bf16 of column tile u at
            // r * 128 + (((2 u + (g >> 1)) ^ (r & 7)) << 4) + 8 ((g & 1) ^ (r >> 3))  (16-byte chunks XOR-swizzled by the row, the two
            // 8-byte halves of a chunk swapped for rows 8-15: conflict-free for these stores and for both read-back layouts)
            const int r = lane_e & 15, g = lane_e >> 4;
            const unsigned a0 = stg_base + (unsigned)(r * 128 + ((((g >> 1) ^ (r & 1)) | (r & 6)) << 4) + 8 * ((g & 1) ^ (r >> 3)));
#pragma unroll
            for (int u = 0; u < NTW; ++u)
                asm volatile("ds_write_b64 %0, %1" ::"v"(a0 ^ (unsigned)(u << 5)), "v"(u32x2_t{pk[I][u][0], pk[I][u][1]}) : "memory");
        }
        if (EPI == P8_GATE) {
            // f32 output: 4 columns per lane (16 lanes = one 256-byte row piece per access), 2 x 4 rows
            const int c4 = lane_e & 15, rq = lane_e >> 4;
            const bool cok = 4 * c4 < WN && pn0 + wc * WN + 4 * c4 < e.N;
            const f32x4 gx = op_x;
            const f32x4 ry[2] = {op_y0, op_y1};
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                const int rs = 8 * HH + 4 * sub + rq;                     // row within the staged tile
                const unsigned ra = stg_base + (unsigned)(rs * 128 + (((c4 >> 1) ^ (rs & 7)) << 4) + 8 * ((c4 & 1) ^ HH));
                u32x2_t w;
                asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(w) : "v"(ra) : "memory");
                __builtin_amdgcn_sched_barrier(0);
                const int row = 16 * I + rs;
                const bool ok = PD_DBG != 4 && cok && row < mrem;
                const unsigned loc = (unsigned)((wr * 64 + row) * e.ldc + wc * WN + 4 * c4);
                if (!e.nt_off || e.nt_aux) __builtin_amdgcn_raw_buffer_store_b64(w, rs_aux, ok ? 2u * loc : EPI_OOB, 0, 2);
                else __builtin_amdgcn_raw_buffer_store_b64(w, rs_aux, ok ? 2u * loc : EPI_OOB, 0, 0);
                f32x4 v = {pd_lo(w[0]), pd_hi(w[0]), pd_lo(w[1]), pd_hi(w[1])};
                v = v * gx + ry[sub];
                buf_store16(rs_c, ok ? 4u * loc : EPI_OOB, v, !e.nt_off);
            }
            return;
        }
        // bf16 outputs: 8 columns per lane, 8 rows per pass
        const int rd_row = lane_e >> 3, c8 = lane_e & 7;
        const bool col_ok = 8 * c8 < WN && pn0 + wc * WN + 8 * c8 < e.N;
        const unsigned ra0 = stg_base + (unsigned)(rd_row * 128 + ((c8 ^ rd_row) << 4));
        constexpr int P0 = RPS == 16 ? 0 : HH, P1 = RPS == 16 ? 2 : HH + 1;
#pragma unroll
        for (int p = P0; p < P1; ++p) {
            u32x4_t w;
            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(w) : "v"(ra0 + (unsigned)(p << 10)) : "memory");
            __builtin_amdgcn_sched_barrier(0);
            if (p == 1) w = u32x4_t{w[2], w[3], w[0], w[1]};             // rows 8-15: the 8-byte halves of a chunk are stored swapped
            const int row = 16 * I + 8 * p + rd_row;
            const bool ok = PD_DBG != 4 && col_ok && row < mrem;
            const unsigned loc = (unsigned)((wr * 64 + row) * e.ldc + wc * WN + 8 * c8);
            if (EPI == P8_GELU) {
                buf_store16(rs_aux, ok ? 2u * loc : EPI_OOB, __builtin_bit_cast(f32x4, w), !e.nt_off || e.nt_aux);
#pragma unroll
                for (int j = 0; j < 4; ++j) w[j] = pd_pack2(gelu_tanh_fast(pd_lo(w[j])), gelu_tanh_fast(pd_hi(w[j])));
            } else if (EPI == P8_DGELU) {
                const u32x4_t h = __builtin_bit_cast(u32x4_t, op_x);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    w[j] = pd_pack2(pd_lo(w[j]) * gelu_tanh_grad_fast(pd_lo(h[j])), pd_hi(w[j]) * gelu_tanh_grad_fast(pd_hi(h[j])));
            }
            buf_store16(rs_c, ok ? 2u * loc : EPI_OOB, __builtin_bit_cast(f32x4, w), !e.nt_off);
            if (EK::may_colsum) {
                const f32x4 a = {pd_lo(w[0]), pd_hi(w[0]), pd_lo(w[1]), pd_hi(w[1])}, b = {pd_lo(w[2]), pd_hi(w[2]), pd_lo(w[3]), pd_hi(w[3])};
                cs0 += ok ? a : zero4;
                cs1 += ok ? b : zero4;
                asm volatile("" : "+v"(cs0), "+v"(cs1));
            }
        }
        if (EK::may_colsum && S == STEPS - 1) {
            if (e.colpart != nullptr) {
                // 8 row groups of the read-back layout: fold lane bits 3, 4, 5 in a fixed order; one partial row per 64 rows
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    cs0[j] += __shfl_xor(cs0[j], 8, 64); cs0[j] += __shfl_xor(cs0[j], 16, 64); cs0[j] += __shfl_xor(cs0[j], 32, 64);
                    cs1[j] += __shfl_xor(cs1[j], 8, 64); cs1[j] += __shfl_xor(cs1[j], 16, 64); cs1[j] += __shfl_xor(cs1[j], 32, 64);
                }
                if (lane_e < 8 && col_ok && mrem > 0) {
                    float* cp = e.colpart + (2 * (int64_t)ptm + wr) * e.N + pn0 + wc * WN + 8 * c8;
                    store4(cp, cs0);
                    store4(cp + 4, cs1);
                }
            }
            cs0 = zero4;
            cs1 = zero4;
        }
    };
    int cstage = 0;
    const int wn0 = wc * WN;
    for (; it_cur < n_items; it_cur += G) {
        int tm, tn;
        decode(it_cur, tm, tn);
        const int64_t m0 = (int64_t)tm * PD_BM, n0 = (int64_t)tn * Cfg::BN;
        if (PD_DBG != 2 && PD_DBG != 1 && EPI != P8_DGELU && e.bias) {     // in flight during the K loop (>= 4 K tiles: covered by its counted waits long before the park)
            const int col = lane < WN ? lane : 0;
            pd_load4<PD_R_BIAS>(pd_uniform(e.bias), (n0 + wn0 + col < e.N) ? 4u * (unsigned)(n0 + wn0 + col) : 0u);
        }
        f32x4 acc[4][NTW];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int u = 0; u < NTW; ++u) acc[i][u] = f32x4{0, 0, 0, 0};
        if (!PD_ONEBAR && wr == 1) __builtin_amdgcn_s_barrier();       // waves 4-7 run half a phase behind waves 0-3
        int lane_k = lane;
        asm volatile("" : "+v"(lane_k));
        // One K tile.  J >= 0: the J-th K tile of an item whose predecessor is parked -- its drain slot J (J < DMAX) runs in phase 2 and
        // the counted waits include the ND operations of every live slot younger than the piece they wait for (slots of K tiles
        // J - 2 .. J); J = -1: the steady K tile, nothing parked or everything drained.  Compile-time, so the K loop carries no
        // branch but its own (the first build selected slot and wait counts by scalar branches inside the loop: every phase
        // paid for ~10 taken branches, 96.7 against 75.5 us on the K = 3072 input gradient with the stores switched off).
        auto ktile = [&](auto jj) __attribute__((always_inline)) {
            constexpr int J = decltype(jj)::value;
            constexpr auto live = [](int j) { return (J >= 0 && j >= 0 && j < DMAX && PD_DBG != 3) ? 1 : 0; };
            constexpr int C1 = live(J - 2) + live(J - 1), C2 = live(J - 1) + live(J);
            const char* st = smem + cstage * Cfg::stage_bytes;
            cstage = cstage == 2 ? 0 : cstage + 1;
            bf16x8 bfr[2][NTW], af[2][2];
            auto load_a = [&](int j) __attribute__((always_inline)) {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int s = 0; s < 2; ++s) af[s][t] = p8_frag<true>(st + j * P8_PART, wr * 32 + 16 * t, s, lane_k);
            };
#define PD_MMA(j)                                                                                                     \
    do {                                                                                                              \
        if (!BKM) {             /* transposed fragments are read by inline asm: the compiler does not wait for them itself */ \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                        \
            __builtin_amdgcn_sched_barrier(0);                                                                        \
        }                                                                                                             \
        __builtin_amdgcn_s_setprio(1);                                                                                \
        _Pragma("unroll") for (int s = 0; s < 2; ++s)                                                                 \
            _Pragma("unroll") for (int t = 0; t < 2; ++t)                                                             \
                _Pragma("unroll") for (int u = 0; u < NTW; ++u)                                                       \
                    acc[2 * (j) + t][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[s][u], af[s][t], acc[2 * (j) + t][u], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);                                                                                \
    } while (0)
            if (PD_ONEBAR) {
                pd_vmwait<LS + C1 * ND>();     // every piece of this K tile's set (its last one was issued two K tiles ago); younger: the next set
                __builtin_amdgcn_s_barrier();
                issue_p1();
                issue_p2();
                bf16x8 af1[2][2];
#pragma unroll
                for (int u = 0; u < NTW; ++u) {
                    const int n = wn0 + 16 * u;
#pragma unroll
                    for (int s = 0; s < 2; ++s) bfr[s][u] = p8_frag<BKM>(st + Cfg::a_bytes + (n >> 6) * P8_PART, n & 63, s, lane_k);
                }
                load_a(0);
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int s = 0; s < 2; ++s) af1[s][t] = p8_frag<true>(st + P8_PART, wr * 32 + 16 * t, s, lane_k);
                if (live(J)) run_slot(std::integral_constant<int, (J >= 0 && J < DMAX ? J : 0)>{}, std::false_type{});
                PD_MMA(0);
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int s = 0; s < 2; ++s) af[s][t] = af1[s][t];
                PD_MMA(1);
                return;
            }
            // ---- phase 1: B fragments of the whole K tile, A rows 0-31 of the wave's 64; pieces B0-B2 of the set two K tiles ahead
#pragma unroll
            for (int u = 0; u < NTW; ++u) {
                const int n = wn0 + 16 * u;
#pragma unroll
                for (int s = 0; s < 2; ++s) bfr[s][u] = p8_frag<BKM>(st + Cfg::a_bytes + (n >> 6) * P8_PART, n & 63, s, lane_k);
            }
            load_a(0);
            issue_p1();
            pd_vmwait<LS + 3 + C1 * ND>();     // A part 1 of this K tile (last piece of its set); younger: the next set + the 3 just issued
            __builtin_amdgcn_s_barrier();
            PD_MMA(0);
            __builtin_amdgcn_s_barrier();
            // ---- phase 2: A rows 32-63; the rest of that set; one drain slot of the parked tile
            load_a(1);
            issue_p2();
            if (live(J)) run_slot(std::integral_constant<int, (J >= 0 && J < DMAX ? J : 0)>{}, std::false_type{});
            pd_vmwait<1 + LS + C2 * ND>();     // B parts + A part 0 of the next K tile; younger: its A part 1 and the set just issued
            __builtin_amdgcn_s_barrier();
            PD_MMA(1);
            __builtin_amdgcn_s_barrier();
        };
        int kt = 0;
        if (parked && PD_DBG != 3) {           // (the host guarantees nk >= DMAX + 2)
            pd_unroll(ktile, std::make_integer_sequence<int, DMAX + 2>{});
            kt = DMAX + 2;
        }
        for (; kt < nk; ++kt) ktile(std::integral_constant<int, -1>{});
        if (!PD_ONEBAR && wr == 0) __builtin_amdgcn_s_barrier();       // level the two wave rows
        // ---- the finished tile is parked (measurement builds with PD_DBG = 3 drain the previous one in the open first) ----
        if (PD_DBG == 3 && parked) {
            pd_unroll([&](auto dd) __attribute__((always_inline)) { run_slot(dd, std::true_type{}); }, std::make_integer_sequence<int, DMAX>{});
            pd_vmwait<0>();
        }
        {
            int lane_p = lane;
            asm volatile("" : "+v"(lane_p));
            const int g = lane_p >> 4;
            const float bv = (PD_DBG != 2 && PD_DBG != 1 && EPI != P8_DGELU && e.bias) ? pd_take4<PD_R_BIAS>() : 0.f;
#pragma unroll
            for (int u = 0; u < NTW; ++u) {
                f32x4 b4 = {0, 0, 0, 0};
                if (EPI != P8_DGELU && e.bias) {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        b4[c] = __int_as_float(__builtin_amdgcn_ds_bpermute(4 * (16 * u + 4 * g + c), __float_as_int(bv)));
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x4 v = acc[i][u] * e.alpha + b4;
                    pk[i][u][0] = pd_pack2(v[0], v[1]);
                    pk[i][u][1] = pd_pack2(v[2], v[3]);
                }
            }
        }
        pm0 = m0; pn0 = n0; ptm = tm;
        {
            const int64_t tile_off = m0 * e.ldc + n0;
            rs_c = EK::out_f32(e) ? epi_rsrc((const float*)e.C + tile_off) : epi_rsrc((const bf16_t*)e.C + tile_off);
            if (EPI == P8_GELU || EPI == P8_GATE) rs_aux = epi_rsrc((const bf16_t*)e.aux_out + tile_off);
            if (EPI == P8_DGELU) rl_x = pd_uniform((const bf16_t*)e.aux_in + tile_off);
            if (EPI == P8_GATE) {
                rl_x = pd_uniform(e.gate);
                rl_y = pd_uniform((const float*)e.resid + tile_off);
                const unsigned r0 = (unsigned)(m0 + wr * 64);
                g_smp = r0 / (unsigned)e.rpb;
                g_rin = r0 % (unsigned)e.rpb;
            }
        }
        parked = PD_DBG != 1;
    }
    // the last tile of this workgroup: nothing left to hide it under
    if (parked) pd_unroll([&](auto dd) __attribute__((always_inline)) { run_slot(dd, std::true_type{}); }, std::make_integer_sequence<int, DMAX>{});
    pd_vmwait<0>();        // the stream's trailing out-of-range pieces still target this workgroup's LDS
}

template <bool BKM, int NTW, int EPI>
static void pd_launch_one(const bf16_t* a, int64_t lda, const bf16_t* b, int64_t ldb, int nk, int tiles_m, int tiles_n, int grid,
                          const EpiDev& e, hipStream_t s) {
    static bool attr_done = false;
    const int lds = PdCfg<NTW>::lds_bytes;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)gemm_pd_kernel<BKM, NTW, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_done = true;
    }
    gemm_pd_kernel<BKM, NTW, EPI><<<grid, 512, lds, s>>>(a, lda, b, ldb, nk, tiles_m, tiles_n, e);
}
