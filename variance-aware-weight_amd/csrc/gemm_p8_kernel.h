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
#pragma once
#include <stdlib.h>

#include "gemm_epi.h"

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

// -DP8_TIMING (tools/p8_timing.py, never in the product library): wave 0 and wave 4 of workgroup 0 accumulate s_memtime
// deltas between the five points of a phase and leave them in the first bytes of their output rows.
#ifdef P8_TIMING
#define P8_TS(v) const uint64_t v = __builtin_readcyclecounter()
#define P8_TACC(tA, tB, tC, tD, tE)                                                                       \
    do { tacc[0] += tB - tA; tacc[1] += tC - tB; tacc[2] += tD - tC; tacc[3] += tE - tD; tacc[5] += 1;     \
         if (tprev) tacc[4] += tA - tprev; tprev = tE; } while (0)
#else
#define P8_TS(v) do { } while (0)
#define P8_TACC(tA, tB, tC, tD, tE) do { } while (0)
#endif

#ifndef P8_LATE_AT
#define P8_LATE_AT 2
#endif
// P8_PH2 = 1: the plain k-major-A launches (forward / input gradients of the Linear layers) run TWO phases of 2 x 16 NTW / 4 MFMAs
// per K tile instead of four of half that: the same DMA stream and LDS images, half the barriers and waits per MFMA
#ifndef P8_PH2
#define P8_PH2 0
#endif
// steps the epilogue's operand loads (residual, GELU' argument, gate) run ahead of their use
// (0 = by epilogue kind: 3 for the GELU' input gradient, whose one 16-byte operand per step is the cheapest to hold, 2 otherwise.
//  Measured on the DiT-B/4 launches, one box, interleaved: depth 1 -> 2 -> 3: GELU' input gradient 111.0 -> 104.7 -> 102.4 us,
//  gated fc2 forward 99.4 -> 99.7 -> 97.0, step 13.42 -> 13.33 -> 13.37 ms; depth 3 spills 8 bytes in the 256-column gated kernel)
#ifndef P8_EPI_PD
#define P8_EPI_PD 0
#endif
// 1: the gated-residual epilogue (f32 in, f32 out) works on 4 columns per lane (see the epilogue).  Against the 8-column layout,
// one box, interleaved: DiT-B/4 proj 53.7 -> 50.2 us, fc2 97.4 -> 94.0; DiT-XL/2 fp8 proj 132.4 -> 115.7, fc2 230.4 -> 217.4
// (step 98.5 -> 97.5 ms)
#ifndef P8_GATE_R4
#define P8_GATE_R4 1
#endif
// 1: the same 4-column lanes for the raw f32 slabs of K-split launches (conv weight gradients, half-empty input gradients).  Measured
// on the UNet_64 conv weight gradients (tools/conv_bench.py, interleaved): +-1 %, i.e. nothing -- the slabs are re-read at once and
// live in L2 / MALL; off.
#ifndef P8_SLAB_R4
#define P8_SLAB_R4 0
#endif
// 1: the DMA stream never "ends": past its last item it keeps issuing pieces at out-of-range offsets (zeros into a stage nobody
// reads), so the K loop needs neither the per-piece `if (iss_done)` nor the two forms of every counted wait -- 12 scalar branches
// per K tile gone (lesson of gemm_pd_kernel.h, where such branches cost 25 %)
#ifndef P8_NOBR
#define P8_NOBR 1
#endif
#ifndef P8_NOBR_CONV3
#define P8_NOBR_CONV3 1
#endif
#define P8_BM 256
#define P8_PART 8192
#define P8_EPI_BYTES 32768

// ---- implicit-GEMM conv3x3 (stride 1, pad 1, NHWC activations), template parameter CONV (same arithmetic as the CONV modes
// of gemm_bf16_kernel in gemm.hip; reference ResBlock / stem / head convs, models/unet.py:182-213,492,625):
//   1 forward         A[m][k=(tap,ci)] = x[pix(m)+s(tap)][ci] (gathered, k-major)   B = W[co][(tap,ci)]            (k-major)
//   2 input gradient  A[m][k=(tap,co)] = dy[pix(m)+s(tap)][co] (gathered, k-major)  B[k][n=ci] = W[co][8-tap][ci]  (mn-major window)
//   3 weight gradient, TRANSPOSED: dW^T[(tap,ci)][co] = patches^T . dy
//                     A[k=pix][m=(tap,ci)] = x[pix+s(tap)][ci] (gathered, mn-major)  B = dy [pix][co]               (mn-major)
//     (9*Ci rows fill 256-row tiles far better than Co = 192 k would; the slab reduce transposes back to [co][(tap,ci)])
// The patch matrix never exists: the per-lane SOURCE address of each 16-byte DMA chunk is computed from (pixel, tap,
// channel); padding taps pass an out-of-range buffer offset and read zeros.  K tiles never straddle a tap (channel counts are multiples of 64).
// K order of modes 1 and 2: (64-channel block, tap) with the TAP fastest -- nine consecutive K tiles re-read the same
// 64-channel slab of the tile's pixel neighbourhood (386 pixels x 128 B = 49 KB per workgroup, which the XCD's L2 holds for all
// its workgroups), where the (tap, channel) order of the weight layout walks the whole neighbourhood (Ci x 772 B: 150-600 KB)
// once per tap and thrashes the L2.  The weights need no relayout: a K tile is still 64 contiguous elements of a W row.
struct P8Conv {
    int H, W, Ci, Co;
};

__device__ __forceinline__ void p8_dma16(__amdgpu_buffer_rsrc_t rs, char* lds_dst, unsigned voff, unsigned soff) {
#ifdef P8_ABLATE_DMA       // measurement builds only (tools/p8_timing.py): the main loop without its global -> LDS traffic
    asm volatile("" :: "v"(voff), "s"(soff));
#else
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)lds_dst, 16, voff, soff, 0, 0);
#endif
}

typedef int i32x8_t __attribute__((ext_vector_type(8)));
typedef int i32x4_t __attribute__((ext_vector_type(4)));
// two 16-byte fragment reads -> the 32-byte operand of the fp8 MFMA (k = 32 (l >> 4) + 0..31 of row l & 15)
__device__ __forceinline__ i32x8_t p8_cat(bf16x8 lo, bf16x8 hi) {
    const i32x4_t a = __builtin_bit_cast(i32x4_t, lo), b = __builtin_bit_cast(i32x4_t, hi);
    return i32x8_t{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}

template <int NTW> struct P8Cfg {
    static constexpr int BN = 64 * NTW;              // 4 waves x NTW MFMA tiles of 16 columns
    static constexpr int WN = 16 * NTW;
    static constexpr int a_bytes = 4 * P8_PART;
    static constexpr int stage_bytes = (4 + NTW) * P8_PART;
    static constexpr int lds_bytes = 2 * stage_bytes + P8_EPI_BYTES;
};

__device__ __forceinline__ int p8_mn_swz(int k) { return (((k >> 1) & 1) | (((k >> 3) & 1) << 1)) << 1; }

// Per-lane BYTE offset (from the tile's first element at the current K position) of the 16 bytes this lane's LDS-DMA piece
// `wid` of part `p` fetches.  is_a: the A-part row set (two 32-row runs 128 apart), else 64 p + row.  ES = bytes per element.
// F8 (1-byte elements, a K tile = 128 of them = the same 128-byte rows): the scaled fp8 MFMA 16x16x128 wants 32 consecutive k
// per lane, k = 32 (l >> 4) + j, where the bf16 one wants two 16-byte pieces k = 16 (l >> 4) .. and 64 + 16 (l >> 4) ..; with the
// row's 16-byte chunks laid down in the order 0,2,4,6,1,3,5,7 the two fragment reads of the bf16 kernel (chunks g and 4 + g)
// return exactly bytes [32 g, 32 g + 16) and [32 g + 16, 32 g + 32): same LDS image, same reads, same swizzle.
template <bool KMAJOR, int ES = 2, bool F8 = false>
__device__ __forceinline__ unsigned p8_src_off(bool is_a, int p, int wid, int lane, int64_t ld, int valid) {
    const int r = 8 * wid + (lane >> 3);          // row (k-major) or k (mn-major) within the part
    if (KMAJOR) {
        int chunk = (lane & 7) ^ ((r >> 1) & 7);  // position of the chunk in the LDS row
        if (F8) chunk = chunk < 4 ? 2 * chunk : 2 * (chunk - 4) + 1;     // ... and the 16 source bytes that belong there
        int R = is_a ? (r < 32 ? 32 * p + r : 96 + 32 * p + r) : 64 * p + r;
        R = R < valid ? R : valid - 1;            // LDS-DMA cannot zero-fill: rows beyond the edge mirror a valid row
        return (unsigned)(R * ld * ES) + chunk * 16;
    } else {
        const int chunk = (lane & 7) ^ p8_mn_swz(r);
        int col = is_a ? (chunk < 4 ? 32 * p + 8 * chunk : 96 + 32 * p + 8 * chunk) : 64 * p + 8 * chunk;
        col = col < valid ? col : 0;
        return (unsigned)((r * ld + col) * ES);
    }
}

// 16 (rows r16 .. r16+15 of the part) x 32 (k sub-step s) operand fragment.
template <bool KMAJOR>
__device__ __forceinline__ bf16x8 p8_frag(const char* part, int r16, int s, int lane) {
    if (KMAJOR) {
        const int row = r16 + (lane & 15);
        const int chunk = (4 * s + (lane >> 4)) ^ ((row >> 1) & 7);
#ifdef P8_ABLATE_FRAG      // measurement builds only: no LDS fragment reads (operands = address bits)
        const float f = (float)(row + chunk);
        return bf16x8{(bf16_t)f, (bf16_t)f, (bf16_t)f, (bf16_t)f, (bf16_t)f, (bf16_t)f, (bf16_t)f, (bf16_t)f};
#endif
        return *reinterpret_cast<const bf16x8*>(part + row * 128 + (chunk << 4));
    } else {
        const int li = lane & 15, q = li >> 2, p = li & 3;
        const int kb = 32 * s + 8 * (lane >> 4) + q;
        const int ch = ((r16 >> 3) + (p >> 1)) ^ p8_mn_swz(kb);
        const char* a0 = part + kb * 128 + (ch << 4) + 8 * (p & 1);
        // inline asm, not __builtin_amdgcn_ds_read_tr16_b64: the compiler orders the builtin behind every pending LDS-DMA with
        // s_waitcnt vmcnt(0) (it may alias the DMA's LDS writes), which drains the whole prefetch ring in every phase.  The
        // caller waits with an explicit s_waitcnt lgkmcnt(0) after the phase barrier (P8_LGKM_FENCE) before the MFMAs.
        const unsigned addr = (unsigned)(uintptr_t)(lds_ptr_t)a0;
        bf16x4 lo, hi;
        asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:512"      // k + 4: same swizzle
                     : "=&v"(lo), "=&v"(hi)
                     : "v"(addr)
                     : "memory");
        bf16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return r;
    }
}

// ---- grouped mode (GRP): one launch walks the tiles of MANY weight-gradient problems dW_p = dy_p^T x_p that share K (the
// tokens of the batch).  Whole tiles first (no K split: the workgroups that run concurrently work on neighbouring tiles of
// the same problem at the same K position, so every operand panel is fetched once for all of them), then the tiles of the
// last, partial round of workgroups, K-split into slabs that p8_group_fixup_kernel folds in a fixed order.
struct P8Prob {
    const bf16_t* a;        // dy  [K][M]
    const bf16_t* b;        // x   [K][N]
    float* c;               // dW  [M][N] f32
    int64_t lda, ldb, ldc;
    int M, N, tiles_n, tile0;   // tile0: index of this problem's first tile in the launch's tile list
    float alpha;                // scales the accumulator
    int pad_;
    const float *scale_a, *scale_b;   // fp8 operands: device scalars (per-tensor dequantisation scales) multiplied into alpha, or NULL
};
struct P8Group {
    int n_prob, t_full, t_rem, n_split;   // items: tiles [0, t_full) whole; tiles [t_full, t_full + t_rem) in n_split K ranges
    float* slab;                          // [n_split][t_rem][256][BN] f32
};
// the work item a workgroup (or its DMA stream) is positioned on
struct P8Item {
    const bf16_t *a, *b;      // operand bases (problem level)
    int64_t lda, ldb, m0, n0;
    int M, N, kt0, nk, split, prob, tm, rem;   // split < 0: whole tile (GRP); rem: index among the K-split tiles (GRP)
};

template <bool AK, bool BKM, int NTW, int EPI, bool GRP = false, int CONV = 0, int F8 = 0>   // F8: 0 bf16 | 1 e4m3 x e4m3 | 2 A e5m2, B e4m3
__global__ void __launch_bounds__(512, 2)
gemm_p8_kernel(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ B, int64_t ldb, int nk_total,
               int tiles_m, int tiles_n, int n_split, EpiDev e, int team_delay, const P8Prob* __restrict__ probs = nullptr,
               P8Group grp = P8Group{}, P8Conv cg = P8Conv{}) {
    static_assert(CONV == 0 || !GRP, "grouped launches are plain GEMMs");
    static_assert(!F8 || (AK && BKM && CONV == 0), "fp8 operands: both k-major (the callers keep transposed fp8 copies)");
    constexpr int ES = F8 ? 1 : 2;                 // bytes per operand element
    constexpr int KT = F8 ? 128 : 64;              // operand elements per K tile (128 bytes of a k-major row either way)
    static_assert(CONV == 0 || (CONV == 1 && AK && BKM) || (CONV == 2 && AK && !BKM) || (CONV == 3 && !AK && !BKM), "conv layouts");
    using Cfg = P8Cfg<NTW>;
    constexpr int LS = 4 + NTW;                                  // DMA pieces per wave and K tile
    constexpr bool NOBR = P8_NOBR != 0 && (CONV != 3 || (P8_NOBR_CONV3 != 0 && NTW == 3));   // (the 256-column transposed conv weight gradient spills with it)
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 stages][A parts 0-3 | B parts] | 8 x 4 KiB epilogue images
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wid >> 2, wc = wid & 3;
    // ---- static item list: G resident workgroups; XCD x (workgroups b = x mod 8) owns a contiguous run of every round ----
    const int G = gridDim.x, n_tiles = tiles_m * tiles_n;
    const int n_items = GRP ? grp.t_full + grp.t_rem * grp.n_split : n_tiles * n_split;
    int it_cur;
    {
        const int b = blockIdx.x, x = b & 7, q = G >> 3, r = G & 7;
        it_cur = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
    }
    const bool gm_on = CONV == 0 && e.debug != 3;     // (VAW_GEMM_DEBUG=3: row-major item order, for A/B runs)
    const int nk_per = GRP ? (nk_total + grp.n_split - 1) / grp.n_split : (nk_total + n_split - 1) / n_split;
    auto decode = [&](int it) {
        P8Item t;
        if (GRP) {
            int g;
            if (it < grp.t_full) { g = it; t.split = -1; t.rem = 0; t.kt0 = 0; t.nk = nk_total; }
            else {
                const int j = it - grp.t_full;
                t.split = j / grp.t_rem;
                t.rem = j - t.split * grp.t_rem;
                g = grp.t_full + t.rem;
                t.kt0 = t.split * nk_per;
                t.nk = t.kt0 + nk_per <= nk_total ? nk_per : nk_total - t.kt0;
            }
            int pi = 0;
            while (pi + 1 < grp.n_prob && probs[pi + 1].tile0 <= g) ++pi;      // uniform: scalar loads
            const P8Prob& pr = probs[pi];
            const int tile = g - pr.tile0, tm = tile / pr.tiles_n;
            t.prob = pi; t.tm = tm;
            t.a = pr.a; t.b = pr.b; t.lda = pr.lda; t.ldb = pr.ldb; t.M = pr.M; t.N = pr.N;
            t.m0 = (int64_t)tm * P8_BM;
            t.n0 = (int64_t)(tile - tm * pr.tiles_n) * Cfg::BN;
        } else {
            const int split = it / n_tiles, tile = it - split * n_tiles;
            // wide outputs (>= 8 column tiles): rows in groups of 4, column tile slow within a group, so the 32 consecutive items
            // an XCD works on in a round form a 4 x 8 block of tiles instead of 1-2 rows x all columns -- it then pulls 4 A
            // panels + 8 B panels through its L2 per round, not the whole B operand (DiT-XL fc1: 444 MB fetched for 43 MB of operands)
            int tm, tn;
            if (tiles_n >= 8 && gm_on) {
                const int gsz = 4 * tiles_n, gi = tile / gsz, within = tile - gi * gsz;
                const int rows = tiles_m - 4 * gi < 4 ? tiles_m - 4 * gi : 4;
                tn = within / rows;
                tm = 4 * gi + within - tn * rows;
            } else {
                tm = tile / tiles_n;
                tn = tile - tm * tiles_n;
            }
            t.split = split; t.rem = 0; t.prob = 0; t.tm = tm;
            t.a = A; t.b = B; t.lda = lda; t.ldb = ldb; t.M = (int)e.M; t.N = (int)e.N;
            t.m0 = (int64_t)tm * P8_BM;
            t.n0 = (int64_t)tn * Cfg::BN;
            t.kt0 = split * nk_per;
            t.nk = t.kt0 + nk_per <= nk_total ? nk_per : nk_total - t.kt0;
        }
        return t;
    };

    // ---- the DMA stream (runs ahead of the MFMAs; its own item / K-tile position) ----
    // LDS-DMA by `buffer_load_dwordx4 ... lds`: a scalar buffer resource per operand (base = the item's tile at its first K
    // tile), a 32-bit per-lane byte offset that is fixed for the whole item, and a scalar byte offset that walks K.  No
    // 64-bit per-lane address arithmetic in the loop, and a lane that must read zeros (conv padding) simply passes an
    // out-of-range offset: the hardware returns 0 for it.
    int iss_item = it_cur, iss_kt = 0, iss_nk = 0, iss_stage = 0;
    bool iss_done = iss_item >= n_items;
    __amdgpu_buffer_rsrc_t rs_a = epi_rsrc(A), rs_b = epi_rsrc(B);
    unsigned so_a = 0, so_b = 0, step_a = 0, step_b = 0;
    unsigned off_a[4], off_b[NTW];
    // gather modes.  CONV 1/2: (th, tw) = tap of the K tile the stream is on, c0 = its first channel; vm_a[p] bit t = "tap t of this
    // lane's pixel lies inside the image".  CONV 3: per A piece the pixel (h, w) of this lane's k row, carried from K tile to K
    // tile, and the tap shift of its column.
    const int conv_cin = CONV == 1 ? cg.Ci : cg.Co;
    int iss_th = 0, iss_tw = 0, iss_c0 = 0;
    unsigned vm_a[4];
    int ga_h[4], ga_w[4], ga_dh[4], ga_dw[4];
    const int step_w = CONV == 3 ? 64 % cg.W : 0, step_h = CONV == 3 ? (64 / cg.W) % cg.H : 0;
    auto conv_point = [&]() {                     // CONV 1/2: scalar offsets of the K tile (tap, c0); bases sit (W + 1) pixels early
        so_a = (unsigned)(((iss_th * cg.W + iss_tw) * conv_cin + iss_c0) * 2);
        if (CONV == 1) so_b = (unsigned)(((3 * iss_th + iss_tw) * cg.Ci + iss_c0) * 2);
        if (CONV == 2) so_b = (unsigned)((iss_c0 * (int)ldb + (8 - 3 * iss_th - iss_tw) * cg.Ci) * 2);
    };
    auto iss_open = [&]() {                       // position the stream on the first K tile of item iss_item
        const P8Item t = decode(iss_item);
        iss_nk = t.nk;
        iss_kt = 0;
        const int mvalid = t.M - t.m0 < P8_BM ? (int)(t.M - t.m0) : P8_BM;
        const int nvalid = t.N - t.n0 < Cfg::BN ? (int)(t.N - t.n0) : Cfg::BN;
        const int64_t a_el = AK ? KT : KT * t.lda, b_el = BKM ? KT : KT * t.ldb;      // elements per K tile
        step_a = (unsigned)(a_el * ES);
        step_b = (unsigned)(b_el * ES);
        so_a = so_b = 0;
        if (CONV == 0) {
            rs_a = epi_rsrc((const char*)t.a + ((AK ? t.m0 * t.lda : t.m0) + t.kt0 * a_el) * ES);
            rs_b = epi_rsrc((const char*)t.b + ((BKM ? t.n0 * t.ldb : t.n0) + t.kt0 * b_el) * ES);
#pragma unroll
            for (int p = 0; p < 4; ++p) off_a[p] = p8_src_off<AK, ES, F8 != 0>(true, p, wid, lane, t.lda, mvalid);
        } else if (CONV == 1 || CONV == 2) {
            const int r = 8 * wid + (lane >> 3), chunk = (lane & 7) ^ ((r >> 1) & 7);
#pragma unroll
            for (int p = 0; p < 4; ++p) {         // this lane's row of A part p is one pixel for the whole item
                int R = r < 32 ? 32 * p + r : 96 + 32 * p + r;
                R = R < mvalid ? R : mvalid - 1;
                const int64_t pix = t.m0 + R;
                const int w = (int)(pix % cg.W), h = (int)((pix / cg.W) % cg.H);
                unsigned m = 0;
#pragma unroll
                for (int tp = 0; tp < 9; ++tp)
                    m |= ((unsigned)(h + tp / 3 - 1) < (unsigned)cg.H && (unsigned)(w + tp % 3 - 1) < (unsigned)cg.W) ? (1u << tp) : 0u;
                vm_a[p] = m;
                off_a[p] = (unsigned)((R * conv_cin + chunk * 8) * 2);
            }
            iss_th = (t.kt0 % 9) / 3;             // K tile index = 9 * channel block + tap
            iss_tw = (t.kt0 % 9) % 3;
            iss_c0 = (t.kt0 / 9) * 64;
            rs_a = epi_rsrc(t.a + (t.m0 - (cg.W + 1)) * conv_cin);
            rs_b = epi_rsrc(CONV == 1 ? t.b + t.n0 * t.ldb : t.b + t.n0);
            conv_point();
        } else {                                  // CONV 3: A = patches^T, gathered: column m = (tap, ci), k row = pixel
            const int kr = 8 * wid + (lane >> 3), chunk = (lane & 7) ^ p8_mn_swz(kr);
            const int64_t pix = (int64_t)t.kt0 * 64 + kr;
            const int w0 = (int)(pix % cg.W), h0 = (int)((pix / cg.W) % cg.H);
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int col = (int)t.m0 + (chunk < 4 ? 32 * p + 8 * chunk : 96 + 32 * p + 8 * chunk);
                const int tap = col < t.M ? col / cg.Ci : -1;             // beyond 9*Ci (edge tile): zeros, always
                ga_dh[p] = tap >= 0 ? tap / 3 - 1 : (1 << 20);
                ga_dw[p] = tap >= 0 ? tap % 3 - 1 : 0;
                ga_h[p] = h0;
                ga_w[p] = w0;
                off_a[p] = tap >= 0 ? (unsigned)(((kr + (ga_dh[p] + 1) * cg.W + ga_dw[p] + 1) * cg.Ci + (col - tap * cg.Ci)) * 2) : EPI_OOB;
                if (step_w == 0 && (unsigned)(w0 + ga_dw[p]) >= (unsigned)cg.W) off_a[p] = EPI_OOB;
            }
            rs_a = epi_rsrc(t.a + ((int64_t)t.kt0 * 64 - (cg.W + 1)) * cg.Ci);
            rs_b = epi_rsrc(t.b + t.n0 + t.kt0 * b_el);
            step_a = (unsigned)(64 * cg.Ci * 2);
        }
#pragma unroll
        for (int p = 0; p < NTW; ++p) off_b[p] = p8_src_off<BKM, ES, F8 != 0>(false, p, wid, lane, t.ldb, nvalid);
    };
    if (!iss_done) iss_open();
    auto iss_close = [&]() {                      // P8_NOBR: every later piece reads out of range
#pragma unroll
        for (int p = 0; p < 4; ++p) off_a[p] = EPI_OOB;
#pragma unroll
        for (int p = 0; p < NTW; ++p) off_b[p] = EPI_OOB;
        if (CONV == 1 || CONV == 2) {
#pragma unroll
            for (int p = 0; p < 4; ++p) vm_a[p] = 0u;
        }
        if (CONV == 3) {
#pragma unroll
            for (int p = 0; p < 4; ++p) ga_dh[p] = 1 << 20;
        }
    };
    if (NOBR && iss_done) iss_close();
    // piece c of the stream order [B parts 0..NTW-1, A parts 0..3]
    auto iss_piece = [&](auto cc) {
        constexpr int c = decltype(cc)::value;
        if (!NOBR && iss_done) return;
        char* dst = smem + iss_stage * Cfg::stage_bytes + wid * 1024;
        if (c < NTW) {
            p8_dma16(rs_b, dst + Cfg::a_bytes + c * P8_PART, off_b[c < NTW ? c : 0], so_b);
        } else {
            constexpr int p = c >= NTW ? c - NTW : 0;
            unsigned voff = off_a[p];
            if (CONV == 1 || CONV == 2) {
                voff = ((vm_a[p] >> (3 * iss_th + iss_tw)) & 1u) ? voff : EPI_OOB;
            } else if (CONV == 3) {
                if (step_w == 0) {                // W divides 64 (every UNet level): the column never changes, it was folded
                                                  // into off_a when the item was opened; only the row moves
                    voff = (unsigned)(ga_h[p] + ga_dh[p]) < (unsigned)cg.H ? voff : EPI_OOB;
                } else {
                    const bool in = (unsigned)(ga_h[p] + ga_dh[p]) < (unsigned)cg.H && (unsigned)(ga_w[p] + ga_dw[p]) < (unsigned)cg.W;
                    voff = in ? voff : EPI_OOB;
                    ga_w[p] += step_w;            // the next K tile of this piece: 64 pixels further on
                    if (ga_w[p] >= cg.W) { ga_w[p] -= cg.W; ga_h[p] += 1; }
                }
                ga_h[p] += step_h;
                if (ga_h[p] >= cg.H) ga_h[p] -= cg.H;
            }
            p8_dma16(rs_a, dst + p * P8_PART, voff, so_a);
        }
    };
    auto iss_advance = [&]() {                    // after the last piece of a K tile
        if (!NOBR && iss_done) return;
        if (NOBR && iss_done) { iss_stage ^= 1; return; }
        iss_stage ^= 1;
        if (CONV == 1 || CONV == 2) {
            if (++iss_tw == 3) { iss_tw = 0; if (++iss_th == 3) { iss_th = 0; iss_c0 += 64; } }
        } else {
            so_a += step_a;
            so_b += step_b;
        }
        if (++iss_kt == iss_nk) {
            iss_item += G;
            if (iss_item >= n_items) { iss_done = true; if (NOBR) iss_close(); }
            else iss_open();
        } else if (CONV == 1 || CONV == 2) {
            conv_point();
        }
    };
#define P8_PIECE(c) iss_piece(std::integral_constant<int, (c)>{})
    // issue slots of the four phases (stream positions; NTW = 4: 2,2,2,2; NTW = 3: 2,2,2,1)
    auto issue_ph1 = [&]() { P8_PIECE(4); P8_PIECE(5); };
    auto issue_ph2 = [&]() { P8_PIECE(6); if (LS == 8) P8_PIECE(LS - 1); iss_advance(); };
    auto issue_ph3 = [&]() { P8_PIECE(0); P8_PIECE(1); };
    auto issue_ph4 = [&]() { P8_PIECE(2); P8_PIECE(3); };
    // counted waits: pieces younger than the one the NEXT phase reads (derivation in DESIGN.md §5).
    // LATE (gather modes and mn-major A -- convs and weight gradients): a phase's two pieces are issued between the two halves of
    // its MFMAs instead of before its wait, so their address arithmetic and any issue stall run under MFMAs already in the pipe
    // (UNet_64 step -4.9 %, plain weight gradients +6 %); the wait then counts two pieces fewer.  Plain k-major A keeps the early
    // issue (late placement measured -4 ... -8 % there).
    constexpr bool LATE = CONV != 0 || !AK;
#define P8_WAIT(n_full)                                                                     \
    do {                                                                                    \
        if (!NOBR && iss_done) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              \
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
    if (team_delay > 0 && ((blockIdx.x >> 3) & 1)) {
        // every other workgroup of an XCD starts `team_delay` x 10 ns late (its first K tiles are already in flight): the
        // two teams then reach their HBM-bound epilogues at different times instead of all 256 CUs storing at once
        const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
        while (__builtin_amdgcn_s_memrealtime() - t0 < (uint64_t)team_delay) __builtin_amdgcn_s_sleep(16);
    }
    constexpr bool PH2 = P8_PH2 != 0 && CONV == 0 && AK && F8 == 0 && !GRP;
    if (PH2) P8_WAIT(6); else P8_WAIT(7);          // (PH2's first phase also reads A part 1)
    __builtin_amdgcn_s_barrier();

    int stage = 0;
    const int wn0 = wc * Cfg::WN;
#ifdef P8_TIMING
    uint64_t tacc[6] = {0, 0, 0, 0, 0, 0}, tprev = 0;
#endif
    for (; it_cur < n_items; it_cur += G) {
        const P8Item item = decode(it_cur);
#ifdef P8_TIMING
        tprev = 0;
#endif
        const int split = item.split, tm = item.tm, nk = item.nk;
        const int64_t m0 = item.m0, n0 = item.n0;
        // per-item view of the epilogue descriptor: in grouped mode the output tensor changes from item to item
        EpiDev ei = e;
        if (F8) { /* scales folded in below, once the item's descriptor is final */ }
        if (GRP) {
            const P8Prob& pr = probs[item.prob];
            ei.C = pr.c; ei.ldc = pr.ldc; ei.M = pr.M; ei.N = pr.N; ei.alpha = pr.alpha;
            ei.scale_a = pr.scale_a; ei.scale_b = pr.scale_b;
        }
        if (F8) {
            if (ei.scale_a) ei.alpha *= *ei.scale_a;
            if (ei.scale_b) ei.alpha *= *ei.scale_b;
        }
        if (EPI == P8_GELU_Q || EPI == P8_DGELU_Q) ei.q_inv = 1.f / ei.q_state[0];
        f32x4 acc[8][NTW];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int u = 0; u < NTW; ++u) acc[i][u] = f32x4{0, 0, 0, 0};
        // CONV 3: the bias gradient sum_pixels dy[.][co] rides along as column sums of the B operand, taken by the upper wave
        // row of the first row tile (tm == 0) of every K split; partial sums per split go to e.rowpart [split][N]
        const bool do_colsum_b = CONV == 3 && e.rowpart != nullptr && tm == 0 && wr == 0;
        f32x4 accb[NTW];
#pragma unroll
        for (int u = 0; u < NTW; ++u) accb[u] = f32x4{0, 0, 0, 0};
        const bf16x8 ones8 = {(bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f};
        if (wr == 1) __builtin_amdgcn_s_barrier();       // waves 4-7 run half a phase behind waves 0-3
        int lane_k = lane;
        asm volatile("" : "+v"(lane_k));      // opaque per item: fragment addresses are rebuilt per item, not kept live (and
                                              // spilled) across the epilogue, whose register budget is the tight one
        for (int kt = 0; kt < nk; ++kt) {
            const char* st = smem + stage * Cfg::stage_bytes;
            stage ^= 1;
            bf16x8 bfr[2][NTW], af[2][2];
            auto load_a = [&](int j) {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int s = 0; s < 2; ++s) af[s][t] = p8_frag<AK>(st + j * P8_PART, wr * 32 + 16 * t, s, lane_k);
            };
#define P8_MMA(j, ISSUE)                                                                                              \
    do {                                                                                                              \
        if (!AK || !BKM) {      /* fragments read by inline asm: the compiler does not wait for them itself */          \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                        \
            __builtin_amdgcn_sched_barrier(0);                                                                        \
        }                                                                                                             \
        __builtin_amdgcn_s_setprio(1);                                                                                \
        if (F8) {   /* one scaled fp8 MFMA (K = 128, unit block scales) where the bf16 kernel issues two of K = 32 */   \
            _Pragma("unroll") for (int t = 0; t < 2; ++t) {                                                           \
                if (t == 1) { ISSUE; }                                                                                \
                _Pragma("unroll") for (int u = 0; u < NTW; ++u)                                                       \
                    /* the MFMA's first matrix is our B tile (cbsz: e4m3), its second our A tile (blgp: 0 e4m3 | 1 e5m2) */ \
                    acc[2 * (j) + t][u] = F8 == 2 ? __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(                  \
                        p8_cat(bfr[0][u], bfr[1][u]), p8_cat(af[0][t], af[1][t]), acc[2 * (j) + t][u], 0, 1, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F) \
                                                  : __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(                  \
                        p8_cat(bfr[0][u], bfr[1][u]), p8_cat(af[0][t], af[1][t]), acc[2 * (j) + t][u], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F); \
            }                                                                                                         \
        } else {                                                                                                      \
            _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                                           \
                /* this phase's two DMA pieces go out between the two halves of its MFMAs: an issue stall (the texture \
                   addresser takes 16 cycles per wave-instruction) then overlaps MFMAs already in the pipe */          \
                _Pragma("unroll") for (int t = 0; t < 2; ++t) {                                                       \
                    if (2 * s + t == P8_LATE_AT) { ISSUE; }     /* quarter of the MFMA block after which the pieces go out */ \
                    _Pragma("unroll") for (int u = 0; u < NTW; ++u)                                                   \
                        acc[2 * (j) + t][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[s][u], af[s][t], acc[2 * (j) + t][u], 0, 0, 0); \
                }                                                                                                     \
            }                                                                                                         \
        }                                                                                                             \
        __builtin_amdgcn_s_setprio(0);                                                                                \
        if (F8) {   /* pin the results here: the accumulators are only read after the loop, and LLVM's code sinking   \
                       otherwise moves all four phases' MFMAs (pure register ops) into the loop's last block */       \
            _Pragma("unroll") for (int t = 0; t < 2; ++t)                                                             \
                _Pragma("unroll") for (int u = 0; u < NTW; ++u) asm volatile("" : "+v"(acc[2 * (j) + t][u]));        \
        }                                                                                                             \
    } while (0)
            if constexpr (PH2) {
                // ---- two phases per K tile.  Phase A: B fragments of the whole K tile, A quarters 0 and 1, the stream's pieces
                // NTW .. LS-1 of the NEXT K tile (its A parts); phase B: A quarters 2 and 3, pieces 0 .. 3 of the K tile after
                // (B parts (+ A part 0)).  Counted waits (pieces younger than the last one the next phase reads): end of A -- the
                // whole next set = LS; end of B -- the last two A parts of the next set + the four just issued = 6.
                // Every wave waits for ITS fragment reads before the barrier: the other wave row issues DMA into the parts they
                // came from right behind that barrier (in the four-phase schedule a whole phase lies between).
                bf16x8 af4[2][4];
                auto load_a2 = [&](int h) {
#pragma unroll
                    for (int jq = 0; jq < 2; ++jq)
#pragma unroll
                        for (int t = 0; t < 2; ++t)
#pragma unroll
                            for (int s = 0; s < 2; ++s)
                                af4[s][2 * jq + t] = p8_frag<AK>(st + (2 * h + jq) * P8_PART, wr * 32 + 16 * t, s, lane_k);
                };
#define P8_MMA2(h)                                                                                                    \
    do {                                                                                                              \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                            \
        __builtin_amdgcn_s_setprio(1);                                                                                \
        _Pragma("unroll") for (int s = 0; s < 2; ++s)                                                                 \
            _Pragma("unroll") for (int i4 = 0; i4 < 4; ++i4)                                                          \
                _Pragma("unroll") for (int u = 0; u < NTW; ++u)                                                       \
                    acc[4 * (h) + i4][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[s][u], af4[s][i4], acc[4 * (h) + i4][u], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);                                                                                \
    } while (0)
#pragma unroll
                for (int u = 0; u < NTW; ++u) {
                    const int n = wn0 + 16 * u;
#pragma unroll
                    for (int s = 0; s < 2; ++s) bfr[s][u] = p8_frag<BKM>(st + Cfg::a_bytes + (n >> 6) * P8_PART, n & 63, s, lane_k);
                }
                load_a2(0);
                issue_ph1();
                issue_ph2();
                if (LS == 8) P8_WAIT(8); else P8_WAIT(7);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                P8_MMA2(0);
                __builtin_amdgcn_s_barrier();
                load_a2(1);
                issue_ph3();
                issue_ph4();
                P8_WAIT(6);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                P8_MMA2(1);
                __builtin_amdgcn_s_barrier();
                continue;
            }
            // ---- phase 1: B fragments of the whole K tile + A quarter 0
#pragma unroll
            for (int u = 0; u < NTW; ++u) {
                const int n = wn0 + 16 * u;
#pragma unroll
                for (int s = 0; s < 2; ++s) bfr[s][u] = p8_frag<BKM>(st + Cfg::a_bytes + (n >> 6) * P8_PART, n & 63, s, lane_k);
            }
            P8_TS(t1a);
            load_a(0);
            if (!LATE) issue_ph1();
            P8_TS(t1b);
            if (LATE) P8_WAIT(6); else P8_WAIT(8);
            P8_TS(t1c);
            __builtin_amdgcn_s_barrier();
            P8_TS(t1d);
            P8_MMA(0, if (LATE) issue_ph1());
            if (CONV == 3 && do_colsum_b) {      // column sums of the B tile (dy): B . ones -- every column of the result holds them
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int u = 0; u < NTW; ++u) accb[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[s][u], ones8, accb[u], 0, 0, 0);
            }
            P8_TS(t1e);
            P8_TACC(t1a, t1b, t1c, t1d, t1e);
            __builtin_amdgcn_s_barrier();
            // ---- phase 2
            P8_TS(t2a);
            load_a(1);
            if (!LATE) issue_ph2();
            P8_TS(t2b);
            if (LATE) P8_WAIT(7); else if (LS == 8) P8_WAIT(9); else P8_WAIT(8);
            P8_TS(t2c);
            __builtin_amdgcn_s_barrier();
            P8_TS(t2d);
            P8_MMA(1, if (LATE) issue_ph2());
            P8_TS(t2e);
            P8_TACC(t2a, t2b, t2c, t2d, t2e);
            __builtin_amdgcn_s_barrier();
            // ---- phase 3
            P8_TS(t3a);
            load_a(2);
            if (!LATE) issue_ph3();
            P8_TS(t3b);
            if (LATE) { if (LS == 8) P8_WAIT(8); else P8_WAIT(7); } else { if (LS == 8) P8_WAIT(10); else P8_WAIT(9); }
            P8_TS(t3c);
            __builtin_amdgcn_s_barrier();
            P8_TS(t3d);
            P8_MMA(2, if (LATE) issue_ph3());
            P8_TS(t3e);
            P8_TACC(t3a, t3b, t3c, t3d, t3e);
            __builtin_amdgcn_s_barrier();
            // ---- phase 4
            P8_TS(t4a);
            load_a(3);
            if (!LATE) issue_ph4();
            P8_TS(t4b);
            if (LATE) P8_WAIT(5); else P8_WAIT(7);
            P8_TS(t4c);
            __builtin_amdgcn_s_barrier();
            P8_TS(t4d);
            P8_MMA(3, if (LATE) issue_ph4());
            P8_TS(t4e);
            P8_TACC(t4a, t4b, t4c, t4d, t4e);
            __builtin_amdgcn_s_barrier();
        }
        if (wr == 0) __builtin_amdgcn_s_barrier();       // level the two wave rows: both run their epilogues together

        if (CONV == 3 && do_colsum_b && (lane & 15) == 0) {      // lane l (column 0 of the MFMA result) holds rows 4 (l >> 4) + reg
#pragma unroll
            for (int u = 0; u < NTW; ++u) {
                const int64_t nn = n0 + wn0 + 16 * u + 4 * (lane >> 4);
                if (nn < e.N) store4(e.rowpart + (int64_t)split * e.N + nn, accb[u]);        // N % 8 == 0
            }
        }
        if (ei.debug == 1) {   // measurement only (VAW_GEMM_DEBUG=1): no epilogue; one never-taken store keeps the accumulators alive
            float t = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int u = 0; u < NTW; ++u) t += acc[i][u][0] + acc[i][u][1] + acc[i][u][2] + acc[i][u][3];
            if (t == 12345.678f) ((float*)ei.C)[0] = t;
            continue;
        }
        // ---- epilogue: 8 row tiles of 16 rows through this wave's private LDS image ----
        char* ep = smem + 2 * Cfg::stage_bytes + wid * 4096;
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));      // opaque per item: the epilogue's per-lane addresses are recomputed here instead of
                                              // being hoisted out of the item loop and kept (spilled) across the K loop
        const int wr_row = lane_e & 15, wr_g = lane_e >> 4;             // accumulator layout: row, group of 4 columns
        const int rd_row = lane_e >> 3, rd_c8 = lane_e & 7;             // read-back layout: row within 8, group of 8 columns
        // The image is written and read with inline-asm DS instructions: the compiler orders an ordinary LDS store behind
        // every pending LDS-DMA with s_waitcnt vmcnt(0), which would drain the prefetch of the next item and every global
        // store of the previous step, 16 times per item.  DS operations of one wave execute in order, so the write ->
        // read -> write sequence on this wave-private image needs no further fence.
        // 16-byte chunk c of row r lives at r * 256 + ((c ^ r) << 4): (4u + g) ^ r = ((4u) ^ (r & 12)) + (g ^ (r & 3)).
        const unsigned ep_base = (unsigned)(uintptr_t)(lds_ptr_t)ep;
        const unsigned ep_w = ep_base + wr_row * 256 + ((wr_g ^ (wr_row & 3)) << 4);
        unsigned ep_r[2];
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int rr = pass * 8 + rd_row;
            ep_r[pass] = ep_base + rr * 256 + (((2 * rd_c8) ^ rr) << 4);     // chunk 2c; chunk 2c + 1 is this address ^ 16
        }
        const int64_t n = n0 + wn0 + 8 * rd_c8;
        const bool col_ok = 8 * rd_c8 < Cfg::WN && n < ei.N;             // N % 8 == 0: a group is in or out as a whole
        using EK = EpiKind<EPI>;
        // lanes beyond the matrix edge mirror a valid row / column for their loads and skip their stores
        const int64_t n_ld = col_ok ? n : n0;
        const int64_t m_last = ei.M - 1;
        f32x4 b0 = {0, 0, 0, 0}, b1 = {0, 0, 0, 0};
        if (EPI != P8_SLAB && EPI != P8_DGELU && EPI != P8_DGELU_Q && EPI != P8_WGRAD && ei.bias) {
            b0 = load4(ei.bias + n_ld);
            b1 = load4(ei.bias + n_ld + 4);
            asm volatile("" ::"v"(b0), "v"(b1));   // the compiler waits for the bias HERE, once (this drains the DMA prefetch of
                                                   // the next item, issued 2-4 phases ago), instead of with a vmcnt(0) in
                                                   // every step, which would drain the stores of the step before
        }
        f32x4 s0 = {0, 0, 0, 0}, s1 = {0, 0, 0, 0};
        float qmax = 0.f;                     // fp8 output kinds: max |x| of this lane's share of the tile
        // output buffers based at the tile's first element (m0, n0); loc_col = this lane's column offset in the tile
        const int loc_col = wn0 + 8 * rd_c8;
        const bool c_f32 = EK::out_f32(ei);
        const int64_t tile_off = m0 * ei.ldc + n0;
        // raw partial sums: the whole launch (P8_SLAB: slabs [split][M][N]) or the K-split items of a grouped launch
        // (slabs [split][tile][256][BN], dense tiles)
        const bool to_slab = EPI == P8_SLAB || (GRP && split >= 0);
        const int64_t slab_ld = GRP ? Cfg::BN : ei.N;
        const __amdgpu_buffer_rsrc_t rs_c =
            to_slab ? epi_rsrc(GRP ? grp.slab + ((int64_t)split * grp.t_rem + item.rem) * (P8_BM * Cfg::BN)
                                   : ei.slab + (int64_t)split * ei.M * ei.N + m0 * ei.N + n0)
                    : epi_rsrc(EK::out_q ? (const void*)((const unsigned char*)ei.C + tile_off)
                                         : c_f32 ? (const void*)((const float*)ei.C + tile_off) : (const void*)((const bf16_t*)ei.C + tile_off));
        const __amdgpu_buffer_rsrc_t rs_aux = epi_rsrc(EK::aux_out(ei) ? (const void*)((const bf16_t*)ei.aux_out + tile_off) : (const void*)ei.C);
        if constexpr (EPI == P8_SLAB && P8_SLAB_R4 != 0 && !GRP) {
            // ---- raw f32 partial sums of a K split, FOUR columns per lane: one 16-byte store per lane and row, 16 lanes = one
            // 256-byte row piece of the slab (the 8-column layout writes every other 16 bytes of a row per instruction)
            const int r4 = lane_e >> 4, c4 = lane_e & 15;
            const int64_t n4 = n0 + wn0 + 4 * c4;
            const bool col_ok4 = 4 * c4 < Cfg::WN && n4 < ei.N;
            const int loc_col4 = wn0 + 4 * c4;
            const int64_t mrow4 = m0 + wr * 128 + r4, m_last4 = ei.M - 1;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
#pragma unroll
                for (int u = 0; u < NTW; ++u)
                    asm volatile("ds_write_b128 %0, %1" ::"v"(ep_w + (unsigned)(((4 * u) ^ (wr_row & 12)) << 4)), "v"(acc[i][u]) : "memory");
#pragma unroll
                for (int pass = 0; pass < 4; ++pass) {
                    const int64_t m = mrow4 + 4 * (4 * i + pass);
                    const int rr = 4 * pass + r4;
                    f32x4 v;
                    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(ep_base + (unsigned)(rr * 256 + ((c4 ^ rr) << 4))) : "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    const bool ok = col_ok4 && m <= m_last4;
                    buf_store16(rs_c, ok ? 4u * (unsigned)((m - m0) * slab_ld + loc_col4) : EPI_OOB, v, false);
                }
            }
            continue;
        }
        if constexpr (EPI == P8_GATE && P8_GATE_R4 != 0) {
            // ---- gated-residual epilogue, FOUR columns per lane: the f32 output (and the f32 residual it reads) then moves as
            // whole 256-byte row pieces per 16 lanes -- one 16-byte access per lane and row -- where the 8-column layout below
            // issues two accesses per lane that each touch every other 16 bytes of the row.  32 steps of 4 rows; operands 4 steps ahead.
            const int r4 = lane_e >> 4, c4 = lane_e & 15;
            const int64_t n4 = n0 + wn0 + 4 * c4;
            const bool col_ok4 = 4 * c4 < Cfg::WN && n4 < ei.N;
            const int64_t n4_ld = col_ok4 ? n4 : n0;
            f32x4 bb = {0, 0, 0, 0};
            if (ei.bias) {
                bb = load4(ei.bias + n4_ld);
                asm volatile("" ::"v"(bb));
            }
            const int loc_col4 = wn0 + 4 * c4;
            const int64_t mrow4 = m0 + wr * 128 + r4, m_last4 = ei.M - 1;
            const unsigned rpb4 = (unsigned)ei.rpb;
            constexpr int PD4 = 4;
            struct Op4 { f32x4 g, r; } o4[PD4 + 1];
            unsigned smp4 = 0, rin4 = 0;
            int64_t m_ld4 = mrow4;
            auto ld4 = [&](Op4& d, bool first) {
                int64_t mn;
                if (first) {
                    mn = m_ld4 < m_last4 ? m_ld4 : m_last4;
                    smp4 = (unsigned)mn / rpb4;
                    rin4 = (unsigned)mn % rpb4;
                } else {
                    mn = m_ld4 + 4;
                    m_ld4 = mn;
                    if (mn <= m_last4) {
                        rin4 += 4;
                        if (rin4 >= rpb4) {
                            if (rpb4 >= 4) { rin4 -= rpb4; smp4 += 1; }
                            else { smp4 += rin4 / rpb4; rin4 %= rpb4; }
                        }
                    } else {
                        mn = m_last4;
                        smp4 = (unsigned)mn / rpb4;
                        rin4 = (unsigned)mn % rpb4;
                    }
                }
                d.g = load4(ei.gate + (int64_t)smp4 * ei.gate_ld + n4_ld);
                d.r = load4((const float*)ei.resid + mn * ei.ldc + n4_ld);
            };
#pragma unroll
            for (int d = 0; d < PD4; ++d) ld4(o4[d], d == 0);
            typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int i = 0; i < 8; ++i) {
#pragma unroll
                for (int u = 0; u < NTW; ++u)
                    asm volatile("ds_write_b128 %0, %1" ::"v"(ep_w + (unsigned)(((4 * u) ^ (wr_row & 12)) << 4)), "v"(acc[i][u]) : "memory");
#pragma unroll
                for (int pass = 0; pass < 4; ++pass) {
                    const int k = 4 * i + pass;
                    const int64_t m = mrow4 + 4 * k;
                    if (k + PD4 < 32) ld4(o4[(k + PD4) % (PD4 + 1)], false);
                    const int rr = 4 * pass + r4;
                    f32x4 v;
                    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(ep_base + (unsigned)(rr * 256 + ((c4 ^ rr) << 4))) : "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    const bool ok = col_ok4 && m <= m_last4;
                    const unsigned loc = ok ? (unsigned)((m - m0) * ei.ldc) + (unsigned)loc_col4 : 0u;
                    v = v * ei.alpha + bb;
                    const bf16x4 r = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
                    if (!ei.nt_off || ei.nt_aux) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, r), rs_aux, ok ? 2u * loc : EPI_OOB, 0, 2);
                    else __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, r), rs_aux, ok ? 2u * loc : EPI_OOB, 0, 0);
                    v = f32x4{(float)r[0], (float)r[1], (float)r[2], (float)r[3]};
                    const Op4& o = o4[k % (PD4 + 1)];
                    v = v * o.g + o.r;
                    buf_store16(rs_c, ok ? 4u * loc : EPI_OOB, v, !ei.nt_off);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            continue;
        }
        // rows of this lane, step k = 2 i + pass: m = mrow0 + 8 k; (sample, row in sample) carried along for gate / rowadd
        const int64_t mrow0 = m0 + wr * 128 + rd_row;
        const unsigned rpb = (unsigned)ei.rpb;
        unsigned smp = 0, rin = 0;
        // operand loads run PD steps ahead of the step that uses them (ring of PD + 1 operand sets).  vmcnt retires in order and
        // counts stores: the wait for step k's operands also waits for every store issued before their load, i.e. for the stores
        // of step k - PD - 1 and older -- the depth is the slack the stores get to be acknowledged
        constexpr int PD = P8_EPI_PD > 0 ? P8_EPI_PD : (EPI == P8_DGELU || EPI == P8_DGELU_Q) ? 3 : 2;
        EpiOps ops[PD + 1];
        const bool with_ops = EK::loads && (EK::act2(ei) || EK::gate(ei) || EK::resid(ei) || EK::rowadd(ei));
        int64_t m_ld = mrow0;                // row of the next operand load
        auto load_next = [&](EpiOps& dst, bool first) {
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
                mn = m_last;                 // beyond the edge: any valid row (its result is not stored)
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
        for (int i = 0; i < 8; ++i) {
#pragma unroll
            for (int u = 0; u < NTW; ++u)
                asm volatile("ds_write_b128 %0, %1" ::"v"(ep_w + (unsigned)(((4 * u) ^ (wr_row & 12)) << 4)), "v"(acc[i][u]) : "memory");
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                const int k = 2 * i + pass;
                const int64_t m = mrow0 + 8 * k;
                if (with_ops && k + PD < 16)         // operands of step k + PD: in flight while the steps before it compute and store
                    load_next(ops[(k + PD) % (PD + 1)], false);
                f32x4 v0, v1;
                asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(v0), "=&v"(v1)
                             : "v"(ep_r[pass]), "v"(ep_r[pass] ^ 16u)
                             : "memory");
                __builtin_amdgcn_sched_barrier(0);   // nothing that reads v0 / v1 may move above the wait (rule 18)
                const bool ok = col_ok && m <= m_last;
                const int loc = ok ? (int)((m - m0) * ei.ldc) + loc_col : -1;
                if (to_slab) {
                    // split-K partials are re-read at once by the reduce: default cache policy
                    const unsigned bo = ok ? 4u * (unsigned)((m - m0) * slab_ld + loc_col) : EPI_OOB;
                    buf_store16(rs_c, bo, v0, false);
                    buf_store16(rs_c, ok ? bo + 16u : EPI_OOB, v1, false);
                } else {
                    const int64_t mc = m <= m_last ? m : m_last;
                    epi_apply8<EPI>(ei, rs_c, rs_aux, loc, (const float*)ei.C + mc * ei.ldc + n_ld, v0, v1, b0, b1, ops[k % (PD + 1)], qmax);
                    if (EK::may_colsum) {      // unconditional arithmetic (a run-time condition here makes the compiler keep
                                               // all 16 steps' values alive and sum them at the end: spills)
                        const f32x4 z = {0, 0, 0, 0};
                        s0 += ok ? v0 : z;
                        s1 += ok ? v1 : z;
                        asm volatile("" : "+v"(s0), "+v"(s1));   // pins the adds here (the compiler otherwise sinks the whole chain
                                                                 // into the `if (colsum)` block below and spills its 16 inputs)
                    }
                }
                __builtin_amdgcn_sched_barrier(0);   // keep the 16 unrolled steps apart: hoisting across them spills
            }
        }
        if (EK::out_q) {       // this wave's max |x| -> the tensor's running max (integer atomic max on the bits; look first)
            qmax = wave_max(qmax);
            if (lane_e == 0 && qmax > 0.f && !(qmax != qmax)) {
                unsigned* acc_q = reinterpret_cast<unsigned*>(ei.q_state + 1);
                const unsigned mb = __float_as_uint(qmax);
                if (mb > __hip_atomic_load(acc_q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(acc_q, mb);
            }
        }
        if (EK::colsum(ei)) {
            // 8 row groups of the read-back layout: fold lane bits 3, 4, 5 in a fixed order; one partial row per wave row
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s0[j] += __shfl_xor(s0[j], 8, 64); s0[j] += __shfl_xor(s0[j], 16, 64); s0[j] += __shfl_xor(s0[j], 32, 64);
                s1[j] += __shfl_xor(s1[j], 8, 64); s1[j] += __shfl_xor(s1[j], 16, 64); s1[j] += __shfl_xor(s1[j], 32, 64);
            }
            if (lane_e < 8 && col_ok && m0 + wr * 128 < ei.M) {
                float* cp = ei.colpart + (2 * (int64_t)tm + wr) * ei.N + n;
                store4(cp, s0);
                store4(cp + 4, s1);
            }
        }
    }
    if (NOBR) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the stream's trailing out-of-range pieces still target this workgroup's LDS
#ifdef P8_TIMING
    if (blockIdx.x == 0 && lane == 0 && (wid == 0 || wid == 4)) {
        __builtin_amdgcn_s_waitcnt(0);
        uint64_t* out = reinterpret_cast<uint64_t*>((char*)e.C + (int64_t)(wid == 4 ? 128 : 0) * e.ldc * (e.out_f32 ? 4 : 2));
        for (int i = 0; i < 6; ++i) out[i] = tacc[i];
    }
#endif
}


template <bool AK, bool BKM, int NTW, int EPI, int CONV>
static void p8_launch_conv(const bf16_t* a, int64_t lda, const bf16_t* b, int64_t ldb, int nk, int tiles_m, int tiles_n, int split,
                           int grid, const EpiDev& e, const P8Conv& cg, hipStream_t s) {
    static bool attr_done = false;
    const int lds = P8Cfg<NTW>::lds_bytes;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)gemm_p8_kernel<AK, BKM, NTW, EPI, false, CONV>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_done = true;
    }
    gemm_p8_kernel<AK, BKM, NTW, EPI, false, CONV><<<grid, 512, lds, s>>>(a, lda, b, ldb, nk, tiles_m, tiles_n, split, e, 0, nullptr, P8Group{}, cg);
}

template <int NTW, int EPI, int F8>
static void p8_launch_fp8_one(const void* a, int64_t lda, const void* b, int64_t ldb, int nk, int tiles_m, int tiles_n, int split, int grid,
                              const EpiDev& e, hipStream_t s) {
    static bool attr_done = false;
    const int lds = P8Cfg<NTW>::lds_bytes;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)gemm_p8_kernel<true, true, NTW, EPI, false, 0, F8>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_done = true;
    }
    gemm_p8_kernel<true, true, NTW, EPI, false, 0, F8><<<grid, 512, lds, s>>>((const bf16_t*)a, lda, (const bf16_t*)b, ldb, nk, tiles_m, tiles_n,
                                                                               split, e, 0, nullptr, P8Group{}, P8Conv{});
}

template <bool AK, bool BKM, int NTW, int EPI>
static void p8_launch_one(const bf16_t* a, int64_t lda, const bf16_t* b, int64_t ldb, int nk, int tiles_m, int tiles_n, int split,
                          int grid, const EpiDev& e, hipStream_t s, int team_delay) {
    static bool attr_done = false;
    const int lds = P8Cfg<NTW>::lds_bytes;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)gemm_p8_kernel<AK, BKM, NTW, EPI, false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_done = true;
    }
    gemm_p8_kernel<AK, BKM, NTW, EPI, false, 0><<<grid, 512, lds, s>>>(a, lda, b, ldb, nk, tiles_m, tiles_n, split, e, team_delay);
}

template <int NTW, int F8 = 0>
static void p8_launch_group(int nk, int grid, const EpiDev& e, const P8Prob* probs_dev, const P8Group& grp, hipStream_t s) {
    // bf16: dy and x as stored ([K][M], [K][N]: both mn-major); fp8: the callers' transposed copies ([M][K], [N][K]: both k-major)
    static bool attr_done = false;
    const int lds = P8Cfg<NTW>::lds_bytes;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)gemm_p8_kernel<F8 != 0, F8 != 0, NTW, P8_WGRAD, true, 0, F8>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_done = true;
    }
    gemm_p8_kernel<F8 != 0, F8 != 0, NTW, P8_WGRAD, true, 0, F8><<<grid, 512, lds, s>>>(nullptr, 0, nullptr, 0, nk, 0, 0, 1, e, 0, probs_dev, grp, P8Conv{});
}

// C_tile = beta * C_tile + sum_s slab[s][r] (fixed order) for the K-split tiles r of a grouped launch: one workgroup per
// (tile, 32-row band).
template <int BN>
__global__ void p8_group_fixup_kernel(const P8Prob* __restrict__ probs, P8Group grp, float beta) {
    const int r = blockIdx.x >> 3, band = blockIdx.x & 7, g = grp.t_full + r;
    int pi = 0;
    while (pi + 1 < grp.n_prob && probs[pi + 1].tile0 <= g) ++pi;
    const P8Prob pr = probs[pi];
    const int tile = g - pr.tile0, tm = tile / pr.tiles_n, tn = tile - tm * pr.tiles_n;
    const int64_t m0 = (int64_t)tm * P8_BM + 32 * band, n0 = (int64_t)tn * BN;
    constexpr int C4 = BN / 4;
    for (int i = threadIdx.x; i < 32 * C4; i += blockDim.x) {
        const int row = i / C4, c4 = (i - row * C4) * 4;
        const int64_t m = m0 + row, n = n0 + c4;
        if (m >= pr.M || n >= pr.N) continue;                       // N % 8 == 0: a group of 4 is in or out as a whole
        f32x4 acc = {0, 0, 0, 0};
        for (int sp = 0; sp < grp.n_split; ++sp)
            acc += load4(grp.slab + ((int64_t)sp * grp.t_rem + r) * (P8_BM * BN) + (int64_t)(32 * band + row) * BN + c4);
        float* c = pr.c + m * pr.ldc + n;
        acc *= pr.alpha * (pr.scale_a ? *pr.scale_a : 1.f) * (pr.scale_b ? *pr.scale_b : 1.f);
        if (beta != 0.f) acc += beta * load4(c);
        store4(c, acc);
    }
}

// arguments of one launch, as the per-layout translation units receive them
struct P8Launch {
    const bf16_t *a, *b;
    int64_t lda, ldb;
    int nk, tiles_m, tiles_n, split, grid, ntw, epi, team_delay;
};
#define P8_CASE(AKv, BKv, EPIv)                                                                                                  \
    case EPIv:                                                                                                                  \
        if (L.ntw == 4) p8_launch_one<AKv, BKv, 4, EPIv>(L.a, L.lda, L.b, L.ldb, L.nk, L.tiles_m, L.tiles_n, L.split, L.grid, e, s, L.team_delay); \
        else p8_launch_one<AKv, BKv, 3, EPIv>(L.a, L.lda, L.b, L.ldb, L.nk, L.tiles_m, L.tiles_n, L.split, L.grid, e, s, L.team_delay);           \
        break
