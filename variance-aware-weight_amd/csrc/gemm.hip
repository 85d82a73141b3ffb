// Dense contraction kernels (the >97 % of DiT FLOPs):  C[M,N] = epilogue(alpha * op(A) . op(B)).
//
//   gemm_bf16_kernel   bf16 operands, v_mfma_f32_16x16x32_bf16, 128x128x64 block tile, 4 waves (2x2) of
//                      64x64, both operand tiles brought in by LDS-DMA (global_load_lds_dwordx4) into a
//                      double-buffered, XOR-swizzled LDS image (swizzle on the per-lane SOURCE address;
//                      the LDS side of an LDS-DMA is lane-linear), one barrier per K step.
//                      k-major operands ([rows][K]) are read with ds_read_b128, mn-major operands
//                      ([K][rows]: dgrad's W, wgrad's dY and X) with the ds_read_b64_tr_b16 hardware
//                      transpose, so forward, dgrad and wgrad run the same loop with no transposed copies.
//   gemm_generic_kernel any shape / alignment / dtype: register-staged, f32 LDS image, exact-f32
//                      v_mfma_f32_16x16x4_f32.  Serves the f32 parity mode and the odd shapes
//                      (N = p*p*C = 64 head, tiny test models).
//
// MFMA operand maps used (cdna_hip_programming.md §3):
//   16x16x32 bf16: lane l holds A[row l&15][k = 8(l>>4)+j], B[k = 8(l>>4)+j][col l&15], j=0..7
//   16x16x4  f32 : lane l holds A[row l&15][k = l>>4],       B[k = l>>4][col l&15]
//   C/D (both)   : col = l&15, row = 4(l>>4) + reg
#include <stdlib.h>

#include "common.h"
#include <type_traits>

#include "gemm_epi.h"

// =============================================================================================
// Fast path: bf16, M%128==0, N%128==0, K%64==0, 16-byte aligned rows.
// Template BKT = K depth of one pipeline stage:
//   64: 2 x 32 KiB of operand tiles, 2 workgroups per CU -- long-K launches (weight gradients)
//   32: 2 x 16 KiB, epilogue staged in two 64-row halves (33 KiB) -> up to 4 workgroups per CU, so that the
//       HBM-bound epilogue of one workgroup overlaps the MFMA loop of the others -- short-K launches (K = 768)
// =============================================================================================
#define BM 128
#define BN 128
#define BK 64                       // granularity the dispatcher requires of K
#define CS_LD 132                   // f32 row stride of the epilogue staging image (128 + 4: conflict-free)
#define CS_BYTES (64 * CS_LD * 4)   // one 64-row half: 33,792 B
#define CP_BYTES (4 * 128 * 4)      // column-sum scratch

template <int BKT> struct FastCfg {
    static constexpr int tile_bytes = 128 * BKT * 2;
    static constexpr int stage_bytes = 2 * tile_bytes;
    static constexpr int main_bytes = 2 * stage_bytes;
    static constexpr int lds_bytes = (main_bytes > CS_BYTES ? main_bytes : CS_BYTES) + CP_BYTES;
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

// k-major image [128 rows][BKT/8 chunks of 16 B].  The XOR keeps every 16-lane group of a ds_read_b128 of the
// 16x16x32 fragment (row = l&15, chunk = 4s + (l>>4)) on 16 distinct 16-byte slots of the 256-byte bank row:
//   BKT=64 (128-B rows): chunk' = chunk ^ ((row>>1)&7);   BKT=32 (64-B rows): chunk' = chunk ^ ((-(row>>2))&3)
template <int BKT>
__device__ __forceinline__ int kmaj_swz(int row) { return BKT == 64 ? ((row >> 1) & 7) : ((-(row >> 2)) & 3); }
template <int BKT>
__device__ __forceinline__ int kmaj_off(int row, int chunk) { return row * (2 * BKT) + ((chunk ^ kmaj_swz<BKT>(row)) << 4); }
// mn-major image: [BKT k-rows][16 chunks of 16 B]; chunk' = chunk ^ (((row&3)<<2)|((row>>2)&3))
//   (image (b) of cdna_hip_programming.md T10: conflict-free ds_read_b64_tr_b16 for the 16x16x32 operand)
__device__ __forceinline__ int mnmaj_off(int row, int chunk) {
    return row * 256 + ((chunk ^ (((row & 3) << 2) | ((row >> 2) & 3))) << 4);
}

// Stage one 128 x BKT (k-major) or BKT x 128 (mn-major) operand tile into LDS by LDS-DMA: BKT/4 wave-instructions
// of 1 KiB, BKT/16 per wave.  g points at the tile's first element; ld = leading dimension in elements.
template <bool KMAJOR, int BKT>
__device__ __forceinline__ void stage_tile(const bf16_t* __restrict__ g, int64_t ld, char* lds_tile, int wid, int lane,
                                           int valid) {
    // `valid` = rows (k-major) or columns (mn-major, multiple of 8) of this 128-wide tile that exist in the matrix.
    // Out-of-range lanes re-read an in-range chunk instead (LDS-DMA cannot zero-fill): the duplicate data only
    // reaches accumulator rows/columns that the epilogue never stores.
    constexpr int PER_WAVE = BKT / 16;
    constexpr int CPR = BKT / 8;             // 16-byte chunks per k-major row
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
        const int inst = wid * PER_WAVE + i;
        const bf16_t* src;
        if (KMAJOR) {
            int row = inst * (64 / CPR) + lane / CPR;
            const int chunk = (lane % CPR) ^ kmaj_swz<BKT>(row);
            row = row < valid ? row : valid - 1;
            src = g + (int64_t)row * ld + chunk * 8;
        } else {
            const int row = inst * 4 + (lane >> 4);
            int chunk = (lane & 15) ^ (((row & 3) << 2) | ((row >> 2) & 3));
            chunk = chunk * 8 < valid ? chunk : 0;
            src = g + (int64_t)row * ld + chunk * 8;
        }
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(lds_tile + inst * 1024), 16, 0, 0);
    }
}

// Fragment of 16 (rows) x 32 (k) for k-substep s of the tile rows r0..r0+15.
template <bool KMAJOR, int BKT>
__device__ __forceinline__ bf16x8 load_frag(const char* lds_tile, int r0, int s, int lane) {
    if (KMAJOR) {
        const int row = r0 + (lane & 15);
        return *reinterpret_cast<const bf16x8*>(lds_tile + kmaj_off<BKT>(row, 4 * s + (lane >> 4)));
    } else {
        const int li = lane & 15, q = li >> 2, p = li & 3;
        const int kb = 32 * s + 8 * (lane >> 4) + q;
        const int ch = (r0 >> 3) + (p >> 1);
        const char* a0 = lds_tile + mnmaj_off(kb, ch) + 8 * (p & 1);
        const char* a1 = lds_tile + mnmaj_off(kb + 4, ch) + 8 * (p & 1);
        bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)a0);
        bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)a1);
        bf16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return r;
    }
}

// ---- implicit-GEMM conv3x3 (stride 1, pad 1) on NHWC activations: the patch matrix is never materialised; the LDS-DMA
// source address of each 16-byte chunk is computed from (pixel, tap, channel) and padding taps read a zero page.
//   CONV 1 (forward)        A[m][k=(tap,ci)]  = x[pix(m)+s(tap)][ci]                      B = W[co][(tap,ci)] (plain, k-major)
//   CONV 2 (input gradient) A[m][k=(tap,co)]  = dy[pix(m)+s(tap)][co]                     B[k][n=ci] = W[co][8-tap][ci]  (mn-major slice)
//   CONV 3 (weight gradient) A = dy^T (plain, mn-major)                                   B[k=m][n=(tap,ci)] = x[pix(m)+s(tap)][ci] (mn-major)
// K tiles never straddle a tap: Ci (CONV 1) / Co (CONV 2) is a multiple of the stage depth.
struct ConvGeom {
    int H, W, Ci, Co;
};
__device__ __attribute__((aligned(64))) const unsigned char vaw_zero_page[64] = {0};

#define KMAJ_READS(KM) ((KM) ? 1 : 2)      /* LDS read instructions per fragment: one b128, or two transposing b64 */
template <bool AK, bool BKM, int BKT, int CONV>
__global__ void __launch_bounds__(256, BKT == 64 ? 2 : 3)
gemm_bf16_kernel(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ B, int64_t ldb, int nk_total,
                 int tiles_n, int n_wg, int n_split, EpiDev e, ConvGeom cg, int xcd_parts) {
    using Cfg = FastCfg<BKT>;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 stages][A tile | B tile]; reused by the epilogue
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // XCD-aware remap: workgroups b, b+8, ... share an XCD (round-robin dispatch) and its 4 MiB L2.
    //   xcd_parts == 0: 2-D grid (tiles, splits); each XCD gets a contiguous run of tiles (bijective for any grid) so
    //     A row-panels are re-read from that XCD's L2.
    //   xcd_parts == 8 / n_split (n_split in {2, 4, 8}; 1-D grid of 8 * ceil(n_wg / xcd_parts) workgroups): XCD x works on
    //     ONE K range (x % n_split) of ONE contiguous part of the tiles (x / n_split), so the workgroups sharing an L2
    //     walk the same K range in step and every operand panel of that range is fetched once for all of them
    //     (split-K weight gradients: measured HBM fetch was 3x the operand bytes with the tile-run mapping).
    int wg, split_idx;
    if (xcd_parts > 0) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, per = (n_wg + xcd_parts - 1) / xcd_parts;
        split_idx = xcd % n_split;
        wg = (xcd / n_split) * per + j;
        if (j >= per || wg >= n_wg) return;      // padding workgroups of the last part (uniform: before any barrier)
    } else {
        const int orig = blockIdx.x, xcd = orig & 7, q = n_wg >> 3, r = n_wg & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
        split_idx = blockIdx.y;
    }
    // Within an XCD's run, walk the tiles in groups of 8 row panels (row fastest): the workgroups resident on
    // an XCD then touch ~8 A panels + ~8 B panels (3 MB at K=768), which stay in its 4 MiB L2.
    int tm, tn;
    {
        const int tiles_m = n_wg / tiles_n, per_group = 8 * tiles_n;   // n_wg = tiles_m * tiles_n exactly
        const int group = wg / per_group, first_m = group * 8;
        const int gsize = tiles_m - first_m < 8 ? tiles_m - first_m : 8;
        const int in_group = wg - group * per_group;
        tm = first_m + in_group % gsize;
        tn = in_group / gsize;
    }
    const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;
    const int64_t a_step = AK ? BKT : (int64_t)BKT * lda;       // step along k: +BKT (k-major) or +BKT*lda
    const int64_t b_step = BKM ? BKT : (int64_t)BKT * ldb;
    // split-K: split_idx owns k-tiles [kt0, kt0 + nk)   (nk_total counts BKT-deep tiles)
    const int nk_per = (nk_total + n_split - 1) / n_split;
    const int kt0 = split_idx * nk_per;
    const int nk = e.debug == 2 ? 1 : (kt0 + nk_per <= nk_total ? nk_per : nk_total - kt0);
    const int wm = (wid >> 1) * 64, wn = (wid & 1) * 64;
    const int mvalid = e.M - m0 < BM ? (int)(e.M - m0) : BM;     // edge tiles: rows / columns that exist
    const int nvalid = e.N - n0 < BN ? (int)(e.N - n0) : BN;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};

    // ---- per-lane constants of the gather modes (pixel coordinates of the rows / columns this lane stages) ----
    constexpr int PW = BKT / 16, CPR = BKT / 8;
    const bf16_t* zero = reinterpret_cast<const bf16_t*>(vaw_zero_page);
    int ga_h[PW], ga_w[PW];
    int64_t ga_pix[PW];
    if (CONV == 1 || CONV == 2) {
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            int row = (wid * PW + i) * (64 / CPR) + lane / CPR;
            row = row < mvalid ? row : mvalid - 1;
            const int64_t p = m0 + row;
            ga_pix[i] = p * (CONV == 1 ? cg.Ci : cg.Co) + (((lane % CPR) ^ kmaj_swz<BKT>((wid * PW + i) * (64 / CPR) + lane / CPR)) << 3);
            ga_w[i] = (int)(p % cg.W);
            ga_h[i] = (int)((p / cg.W) % cg.H);
        }
    }
    // CONV 3: the pixel (h, w) and source offset of the row each lane stages are carried from K tile to K tile
    // (64 pixels further on each time) instead of being divided out of the pixel index at every step.
    int gb_dh[PW], gb_dw[PW], gb_h[PW], gb_w[PW];
    int64_t gb_off[PW];
    const int step_w = CONV == 3 ? BKT % cg.W : 0, step_h = CONV == 3 ? (BKT / cg.W) % cg.H : 0;
    if (CONV == 3) {
        const int nk_per0 = (nk_total + n_split - 1) / n_split;
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            const int row = (wid * PW + i) * 4 + (lane >> 4);
            const int chunk = (lane & 15) ^ (((row & 3) << 2) | ((row >> 2) & 3));
            const int n = (int)n0 + chunk * 8;
            const int tap = n < e.N ? n / cg.Ci : -1;     // -1: column beyond 9*Ci (edge tile) -> zero page
            gb_dh[i] = tap >= 0 ? tap / 3 - 1 : (1 << 20);
            gb_dw[i] = tap >= 0 ? tap % 3 - 1 : 0;
            const int64_t p = (int64_t)split_idx * nk_per0 * BKT + row;
            gb_w[i] = (int)(p % cg.W);
            gb_h[i] = (int)((p / cg.W) % cg.H);
            gb_off[i] = tap >= 0 ? (p + (int64_t)gb_dh[i] * cg.W + gb_dw[i]) * cg.Ci + n % cg.Ci : 0;
        }
    }
    // CONV 1/2: (tap, first channel) of the K tile about to be staged, carried from tile to tile (A and B side alike)
    const int conv_cin = CONV == 1 ? cg.Ci : cg.Co;
    int a_tap = 0, a_c0 = 0, b_tap = 0, b_c0 = 0;
    if (CONV == 1 || CONV == 2) {
        const int kglob = split_idx * ((nk_total + n_split - 1) / n_split) * BKT;
        a_tap = b_tap = kglob / conv_cin;
        a_c0 = b_c0 = kglob - a_tap * conv_cin;
    }
    auto stage_a = [&](int ktile, char* dst) {        // ktile counts BKT-deep tiles from k = 0; consecutive calls only
        if (CONV == 1 || CONV == 2) {
            const int cin = conv_cin;
            const int tap = a_tap, c0 = a_c0;
            a_c0 += BKT;
            if (a_c0 >= cin) { a_c0 = 0; a_tap += 1; }
            const int dh = tap / 3 - 1, dw = tap % 3 - 1;
            const bf16_t* a_tap_base = A + ((int64_t)dh * cg.W + dw) * cin + c0;     // uniform: one scalar add per piece
#pragma unroll
            for (int i = 0; i < PW; ++i) {
                const int inst = wid * PW + i;
                const int hh = ga_h[i] + dh, ww = ga_w[i] + dw;
                const bool in = hh >= 0 && hh < cg.H && ww >= 0 && ww < cg.W;
                const bf16_t* src = in ? a_tap_base + ga_pix[i] : zero;   // ga_pix: element offset of (pixel, swizzled chunk)
                __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(dst + inst * 1024), 16, 0, 0);
            }
        } else {
            stage_tile<AK, BKT>((AK ? A + m0 * lda : A + m0) + ktile * a_step, lda, dst, wid, lane, mvalid);
        }
    };
    auto stage_b = [&](int ktile, char* dst) {
        if (CONV == 2) {                              // W[co][8-tap][ci]: rows = co, row stride 9*Ci, tap picks the column window
            const int tap = b_tap, c0 = b_c0;
            b_c0 += BKT;
            if (b_c0 >= cg.Co) { b_c0 = 0; b_tap += 1; }
            stage_tile<false, BKT>(B + (int64_t)c0 * ldb + (8 - tap) * cg.Ci + n0, ldb, dst, wid, lane, nvalid);
        } else if (CONV == 3) {       // called for consecutive K tiles only: the per-lane state advances by one tile
#pragma unroll
            for (int i = 0; i < PW; ++i) {
                const int inst = wid * PW + i;
                const int hh = gb_h[i] + gb_dh[i], ww = gb_w[i] + gb_dw[i];
                const bool in = hh >= 0 && hh < cg.H && ww >= 0 && ww < cg.W;
                const bf16_t* src = in ? B + gb_off[i] : zero;
                __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(dst + inst * 1024), 16, 0, 0);
                gb_off[i] += (int64_t)BKT * cg.Ci;
                gb_w[i] += step_w;
                if (gb_w[i] >= cg.W) { gb_w[i] -= cg.W; gb_h[i] += 1; }
                gb_h[i] += step_h;
                if (gb_h[i] >= cg.H) gb_h[i] -= cg.H;
            }
        } else {
            stage_tile<BKM, BKT>((BKM ? B + n0 * ldb : B + n0) + ktile * b_step, ldb, dst, wid, lane, nvalid);
        }
    };
    const bool wave_live = wm < mvalid && wn < nvalid;
    constexpr bool CAN_ROWSUM = CONV == 3 || (CONV == 0 && !AK);
    const bool do_rowsum = CAN_ROWSUM && e.rowpart != nullptr && tn == 0 && wn == 0;   // one column of tiles, its two left waves
    f32x4 accr[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    const bf16x8 ones8 = {(bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f};
    stage_a(kt0, smem);
    stage_b(kt0, smem + Cfg::tile_bytes);
    // The K loop exists twice (compile-time flag): the few waves that also take row sums of A (CONV 3 bias gradient) run
    // their own copy, so the common copy's schedule and register use are untouched by it.
    auto k_loop = [&](auto with_rowsum, auto with_preload) {
        constexpr bool RS = decltype(with_rowsum)::value, PRE = decltype(with_preload)::value;
        for (int kt = 0; kt < nk; ++kt) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's LDS-DMA for tile kt has landed
            __syncthreads();                                    // everyone's has; and tile kt-1 is no longer read
            char* cur = smem + (kt & 1) * Cfg::stage_bytes;
            if (kt + 1 < nk) {
                char* nxt = smem + ((kt + 1) & 1) * Cfg::stage_bytes;
                stage_a(kt0 + kt + 1, nxt);
                stage_b(kt0 + kt + 1, nxt + Cfg::tile_bytes);
            }
            if (!wave_live) continue;      // this wave's 64 x 64 quadrant lies wholly outside the matrix (edge tile)
            if (PRE) {
                // every fragment of the K tile is requested before the first MFMA: the LDS latency of the second 32-deep
                // half hides behind the MFMAs of the first instead of being waited for in the open
                bf16x8 afp[BKT / 32][4], bfp[BKT / 32][4];
#pragma unroll
                for (int s = 0; s < BKT / 32; ++s) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) bfp[s][j] = load_frag<BKM, BKT>(cur + Cfg::tile_bytes, wn + 16 * j, s, lane);
#pragma unroll
                    for (int i = 0; i < 4; ++i) afp[s][i] = load_frag<AK, BKT>(cur, wm + 16 * i, s, lane);
                }
#pragma unroll
                for (int s = 0; s < BKT / 32; ++s)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfp[s][j], afp[s][i], acc[i][j], 0, 0, 0);
                // pin the order: all LDS reads of the tile first, then the MFMAs (mask 0x100 = DS read, 0x008 = MFMA)
                __builtin_amdgcn_sched_group_barrier(0x100, (KMAJ_READS(AK) + KMAJ_READS(BKM)) * 4 * (BKT / 32), 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 16 * (BKT / 32), 0);
                continue;
            }
#pragma unroll
            for (int s = 0; s < BKT / 32; ++s) {
                bf16x8 af[4], bfr[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) af[i] = load_frag<AK, BKT>(cur, wm + 16 * i, s, lane);
#pragma unroll
                for (int j = 0; j < 4; ++j) bfr[j] = load_frag<BKM, BKT>(cur + Cfg::tile_bytes, wn + 16 * j, s, lane);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        // operands swapped: the tile is accumulated TRANSPOSED (lane l: row m = l&15, columns 4(l>>4)..+3),
                        // so the epilogue finds 4 consecutive output columns in one register quad
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
                if (RS) {     // ones . A^T: every row of the 16 x 16 result holds the row sums of A
#pragma unroll
                    for (int i = 0; i < 4; ++i) accr[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones8, af[i], accr[i], 0, 0, 0);
                }
            }
        }
    };
    if (CAN_ROWSUM && do_rowsum) k_loop(std::true_type{}, std::false_type{});
    else if (BKT == 32 && CONV == 0 && AK) k_loop(std::false_type{}, std::true_type{});   // measured: +3..8 % on the 32-deep
                                                                                            // input-gradient launches, neutral elsewhere
    else k_loop(std::false_type{}, std::false_type{});
    if (CAN_ROWSUM && do_rowsum && lane < 16) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = wm + 16 * i + lane;
            if (row < mvalid) e.rowpart[(int64_t)split_idx * e.M + m0 + row] = accr[i][0];
        }
    }
    if (e.debug == 1) {   // keep the accumulators alive with one store per wave-quadrant
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        if (t == 12345.678f) ((float*)e.C)[0] = t;
        return;
    }
    // ---- direct epilogue: every lane owns rows (l&15) of its 4 m-tiles and 4 consecutive columns per n-tile ----
    // (measured: a win only for the f32 split-K partials -- 16 B per lane, 64-B row segments; bf16 outputs written 8 B
    //  per lane in 32-B segments ran 1.5-2x slower than the LDS-staged 16-B rows below, so those keep the staging)
    if (e.direct_epi && !e.colpart && n_split > 1) {
        if (!wave_live) return;
        const int g4 = 4 * (lane >> 4), lr = lane & 15;
        float* slab = n_split > 1 ? e.slab + (int64_t)split_idx * e.M * e.N : nullptr;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = wn + 16 * j + g4;
            if (col >= nvalid) continue;
            const f32x4 bj = (e.bias && n_split == 1) ? load4(e.bias + n0 + col) : f32x4{0, 0, 0, 0};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = wm + 16 * i + lr;
                if (row >= mvalid) continue;
                if (n_split > 1) store4(slab + (m0 + row) * e.N + n0 + col, acc[i][j]);
                else epi_row4(e, (unsigned)(m0 + row), n0 + col, acc[i][j], bj);
            }
        }
        return;
    }
    // ---- epilogue in two 64-row halves: accumulators -> f32 staging image in LDS (the operand tiles are dead),
    //      then each thread owns 8 consecutive columns of one row per pass: 16-byte loads and stores ----
    float* cs = reinterpret_cast<float*>(smem);
    const int c8 = (threadIdx.x & 15) * 8, r0 = threadIdx.x >> 4;
    const bool col_ok = c8 < nvalid;             // N % 8 == 0: an 8-column group is in or out as a whole
    f32x4 b0 = {0, 0, 0, 0}, b1 = {0, 0, 0, 0};
    if (e.bias && n_split == 1 && col_ok) {
        b0 = load4(e.bias + n0 + c8);
        b1 = load4(e.bias + n0 + c8 + 4);
    }
    f32x4 s0 = {0, 0, 0, 0}, s1 = {0, 0, 0, 0};
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        __syncthreads();
        if ((wid >> 1) == half) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    store4(cs + (16 * i + (lane & 15)) * CS_LD + wn + 16 * j + 4 * (lane >> 4), acc[i][j]);
        }
        __syncthreads();
        if (n_split > 1) {
            float* slab = e.slab + (int64_t)split_idx * e.M * e.N;
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                const int row = pass * 16 + r0;
                if (!col_ok || 64 * half + row >= mvalid) continue;
                const float* src = cs + row * CS_LD + c8;
                float* dst = slab + (m0 + 64 * half + row) * e.N + n0 + c8;
                store4(dst, load4(src));        // split-K partials are re-read at once by the reduce: keep them cached
                store4(dst + 4, load4(src + 4));
            }
        } else {
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                const int row = pass * 16 + r0;
                if (!col_ok || 64 * half + row >= mvalid) continue;
                const float* src = cs + row * CS_LD + c8;
                f32x4 v0 = load4(src), v1 = load4(src + 4);
                epi_row8(e, (unsigned)(m0 + 64 * half + row), n0 + c8, v0, v1, b0, b1);
                s0 += v0;
                s1 += v1;
            }
        }
    }
    if (e.colpart && n_split == 1) {
        // fold the 16 row-threads of each 8-column group in a fixed order: 4 in-wave (xor 16, 32), then 4 waves via LDS
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s0[j] += __shfl_xor(s0[j], 16, 64); s0[j] += __shfl_xor(s0[j], 32, 64);
            s1[j] += __shfl_xor(s1[j], 16, 64); s1[j] += __shfl_xor(s1[j], 32, 64);
        }
        float* cp = reinterpret_cast<float*>(smem + Cfg::lds_bytes - CP_BYTES);   // [4 waves][128]
        if (lane < 16) {
            store4(cp + wid * 128 + c8, s0);
            store4(cp + wid * 128 + c8 + 4, s1);
        }
        __syncthreads();
        if (threadIdx.x < nvalid)
            e.colpart[(int64_t)tm * e.N + n0 + threadIdx.x] =
                ((cp[threadIdx.x] + cp[128 + threadIdx.x]) + cp[256 + threadIdx.x]) + cp[384 + threadIdx.x];
    }
}

// =============================================================================================
// Large-tile variant: 256 x 256 block tile, 8 waves (2 x 4, each 128 x 64 = 8 x 4 MFMA tiles), ONE workgroup per
// CU, K in 32-deep tiles through a ring of FOUR 32 KiB LDS stages filled by LDS-DMA.
// Why: the ablations in DESIGN.md ("GEMM main loop") show the 128 x 128 kernel is bound by the vector-memory ->
// LDS path (64 B/clk/CU) and by LDS bandwidth, not by MFMA issue: per MFMA cycle it moves 2x the global bytes
// and 1.33x the LDS fragment bytes of this shape.  The loads of K tile t+3 are issued before the MFMAs of tile t and
// only waited for with a COUNTED s_waitcnt vmcnt(8) (4 LDS-DMA instructions per wave per tile, two tiles may stay
// in flight) followed by a raw s_barrier (__syncthreads() would drain vmcnt to 0; cdna_hip_programming.md
// "Pipelining across barriers").
// Each operand tile is two 128-wide images of the kinds above (same swizzles, same fragment loads).
// =============================================================================================
#define BIG_BM 256
#define BIG_BN 256
#define BIG_IMG (128 * 32 * 2)                  // one 128 x 32 image: 8 KiB
#define BIG_STAGE (4 * BIG_IMG)                 // A img0 | A img1 | B img0 | B img1 = 32 KiB
#define BIG_NSTAGE 4
#define BIG_CS_LD 260                           // f32 row stride of the epilogue staging image (256 + 4)
#define BIG_LDS (BIG_NSTAGE * BIG_STAGE + 8 * 256 * 4)   // ring (also the 64 x 260 f32 epilogue image) + column sums

template <bool AK, bool BKM>
__global__ void __launch_bounds__(512, 1)
gemm_bf16_big_kernel(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ B, int64_t ldb, int nk_total,
                     int tiles_n, int n_wg, int n_split, EpiDev e) {
    constexpr int BKT = 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // 0..7
    int wg;
    {
        const int orig = blockIdx.x, xcd = orig & 7, q = n_wg >> 3, r = n_wg & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    int tm, tn;
    {
        const int tiles_m = n_wg / tiles_n, per_group = 4 * tiles_n;      // 4 row panels of 256 per group
        const int group = wg / per_group, first_m = group * 4;
        const int gsize = tiles_m - first_m < 4 ? tiles_m - first_m : 4;
        const int in_group = wg - group * per_group;
        tm = first_m + in_group % gsize;
        tn = in_group / gsize;
    }
    const int64_t m0 = (int64_t)tm * BIG_BM, n0 = (int64_t)tn * BIG_BN;
    const int nk_per = (nk_total + n_split - 1) / n_split;                // nk_total counts 32-deep tiles
    const int kt0 = blockIdx.y * nk_per;
    const int nk = kt0 + nk_per <= nk_total ? nk_per : nk_total - kt0;
    const int mvalid = e.M - m0 < BIG_BM ? (int)(e.M - m0) : BIG_BM;
    const int nvalid = e.N - n0 < BIG_BN ? (int)(e.N - n0) : BIG_BN;
    const int wy = wid >> 2, wx = wid & 3;                                // wave tile: rows 128*wy.., cols 64*wx..

    // ---- staging: wave w fills (2 pieces each) image w>>2 of A and image w>>2 of B as sub-wave w&3 ----
    const int img = wid >> 2, sw = wid & 3;
    int a_valid = mvalid - 128 * img, b_valid = nvalid - 128 * img;
    const int a_img = a_valid > 0 ? img : 0, b_img = b_valid > 0 ? img : 0;   // image wholly outside: mirror image 0
    a_valid = a_valid > 0 ? (a_valid < 128 ? a_valid : 128) : (mvalid < 128 ? mvalid : 128);
    b_valid = b_valid > 0 ? (b_valid < 128 ? b_valid : 128) : (nvalid < 128 ? nvalid : 128);
    const bf16_t* a_base = AK ? A + (m0 + 128 * a_img) * lda : A + m0 + 128 * a_img;
    const bf16_t* b_base = BKM ? B + (n0 + 128 * b_img) * ldb : B + n0 + 128 * b_img;
    const int64_t a_step = AK ? BKT : (int64_t)BKT * lda;
    const int64_t b_step = BKM ? BKT : (int64_t)BKT * ldb;
    auto stage = [&](int ktile, char* dst) {
        stage_tile<AK, BKT>(a_base + ktile * a_step, lda, dst + img * BIG_IMG, sw, lane, a_valid);
        stage_tile<BKM, BKT>(b_base + ktile * b_step, ldb, dst + (2 + img) * BIG_IMG, sw, lane, b_valid);
    };

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};

    stage(kt0, smem);
    if (nk > 1) stage(kt0 + 1, smem + BIG_STAGE);
    if (nk > 2) stage(kt0 + 2, smem + 2 * BIG_STAGE);
    const int a_off = wy * BIG_IMG, b_off = (2 + (wx >> 1)) * BIG_IMG, b_row0 = (wx & 1) * 64;
    for (int kt = 0; kt < nk; ++kt) {
        // tile kt landed; tiles kt+1, kt+2 may still fly (4 LDS-DMA instructions per wave per tile)
        if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();          // every wave's pieces of tile kt are in; nobody still reads tile kt-1
        if (kt + 3 < nk) stage(kt0 + kt + 3, smem + ((kt + 3) & 3) * BIG_STAGE);   // slot of tile kt-1
        const char* cur = smem + (kt & 3) * BIG_STAGE;
        bf16x8 af[8], bfr[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bfr[j] = load_frag<BKM, BKT>(cur + b_off, b_row0 + 16 * j, 0, lane);
#pragma unroll
        for (int i = 0; i < 8; ++i) af[i] = load_frag<AK, BKT>(cur + a_off, 16 * i, 0, lane);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    if (e.debug == 1) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        if (t == 12345.678f) ((float*)e.C)[0] = t;
        return;
    }
    // ---- epilogue: four 64-row quarters through the (now idle) ring; a thread owns 8 consecutive columns ----
    float* cs = reinterpret_cast<float*>(smem);
    const int c8 = (threadIdx.x & 31) * 8, r0 = threadIdx.x >> 5;       // 16 row-threads x 32 column groups
    const bool col_ok = c8 < nvalid;
    f32x4 b0 = {0, 0, 0, 0}, b1 = {0, 0, 0, 0};
    if (e.bias && n_split == 1 && col_ok) {
        b0 = load4(e.bias + n0 + c8);
        b1 = load4(e.bias + n0 + c8 + 4);
    }
    f32x4 s0 = {0, 0, 0, 0}, s1 = {0, 0, 0, 0};
#pragma unroll
    for (int quarter = 0; quarter < 4; ++quarter) {
        __syncthreads();
        if (wy == (quarter >> 1)) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        cs[(16 * i + 4 * (lane >> 4) + r) * BIG_CS_LD + 64 * wx + 16 * j + (lane & 15)] =
                            acc[4 * (quarter & 1) + i][j][r];
        }
        __syncthreads();
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int row = pass * 16 + r0;
            if (!col_ok || 64 * quarter + row >= mvalid) continue;
            const float* src = cs + row * BIG_CS_LD + c8;
            const int64_t m = m0 + 64 * quarter + row;
            if (n_split > 1) {
                float* dst = e.slab + (int64_t)blockIdx.y * e.M * e.N + m * e.N + n0 + c8;
                store4(dst, load4(src));        // split-K partials are re-read at once by the reduce: keep them cached
                store4(dst + 4, load4(src + 4));
            } else {
                f32x4 v0 = load4(src), v1 = load4(src + 4);
                epi_row8(e, (unsigned)m, n0 + c8, v0, v1, b0, b1);
                s0 += v0;
                s1 += v1;
            }
        }
    }
    if (e.colpart && n_split == 1) {
        // 16 row-threads per column group: 2 in-wave (lane bit 5), then the 8 waves through LDS, fixed order
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s0[j] += __shfl_xor(s0[j], 32, 64);
            s1[j] += __shfl_xor(s1[j], 32, 64);
        }
        float* cp = reinterpret_cast<float*>(smem + BIG_NSTAGE * BIG_STAGE);   // [8 waves][256]
        if (lane < 32) {
            store4(cp + wid * 256 + c8, s0);
            store4(cp + wid * 256 + c8 + 4, s1);
        }
        __syncthreads();
        if ((int)threadIdx.x < nvalid && threadIdx.x < 256) {
            float t = 0.f;
#pragma unroll
            for (int w8 = 0; w8 < 8; ++w8) t += cp[w8 * 256 + threadIdx.x];
            e.colpart[(int64_t)tm * e.N + n0 + threadIdx.x] = t;
        }
    }
}

// out = beta*C + alpha * sum_s slab[s], fixed order (deterministic split-K).  N % 4 == 0.
// The LAST workgroup also folds the [S][M] partial row sums of A (bias gradient), when given: one launch fewer.
template <typename TO>
__global__ void splitk_reduce_kernel(const float* __restrict__ slab, int S, int64_t M, int64_t N, int64_t ldc,
                                     void* __restrict__ C, float alpha, float beta, int out_f32,
                                     const float* __restrict__ rowpart = nullptr, float* __restrict__ rowsum_out = nullptr,
                                     float rowsum_beta = 0.f) {
    if (rowpart && blockIdx.x == gridDim.x - 1) {
        for (int64_t m = threadIdx.x; m < M; m += blockDim.x) {
            float t = 0.f;
            for (int sidx = 0; sidx < S; ++sidx) t += rowpart[(int64_t)sidx * M + m];
            rowsum_out[m] = (rowsum_beta != 0.f ? rowsum_beta * rowsum_out[m] : 0.f) + t;
        }
    }
    const int64_t total4 = M * N / 4;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total4; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = (4 * i) / N, n = (4 * i) % N;
        f32x4 acc = {0, 0, 0, 0};
        // eight slabs requested at once, added in slab order (the sum is the sequential one; only the loads overlap: the small
        // outputs of the long-K launches -- adaLN input gradient, patch-embed / final-layer weight gradients -- give this kernel a
        // few hundred threads each walking 64 slabs, 38 us per call in round 3 with one load in flight)
        int sidx = 0;
        for (; sidx + 8 <= S; sidx += 8) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = load4(slab + (int64_t)(sidx + u) * M * N + 4 * i);
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
        for (; sidx < S; ++sidx) acc += load4(slab + (int64_t)sidx * M * N + 4 * i);
        acc *= alpha;
        const int64_t off = m * ldc + n;
        if (out_f32) {
            float* c = (float*)C + off;
            if (beta != 0.f) acc += beta * load4(c);
            store4(c, acc);
        } else {
            store4((TO*)C + off, acc);
        }
    }
}

// =============================================================================================
// Generic path: any shape; operands f32 or bf16 converted to an f32 LDS image [k][m], stride 144.
// =============================================================================================
#define GBM 128
#define GBN 128
#define GBK 16
#define GLD 144   // 128 + 16: lanes l and l+16 (next k) land on disjoint bank halves for ds_read_b32

template <typename T, bool KMAJOR>
__device__ __forceinline__ void generic_stage(const T* __restrict__ g, int64_t ld, int64_t r0, int64_t rows, int64_t k0,
                                              int64_t K, float* tile, int tid) {
#pragma unroll
    for (int i = 0; i < (GBM * GBK) / 256; ++i) {
        const int e = tid + 256 * i;
        int r, k;
        if (KMAJOR) { r = e / GBK; k = e % GBK; } else { k = e / GBM; r = e % GBM; }
        const int64_t gr = r0 + r, gk = k0 + k;
        float v = 0.f;
        if (gr < rows && gk < K) v = to_f32(KMAJOR ? g[gr * ld + gk] : g[gk * ld + gr]);
        tile[k * GLD + r] = v;
    }
}

template <typename T, bool AK, bool BKM>
__global__ void __launch_bounds__(256)
gemm_generic_kernel(const T* __restrict__ A, int64_t lda, const T* __restrict__ B, int64_t ldb, int64_t K, int64_t kchunk,
                    EpiDev e) {
    __shared__ float As[GBK * GLD];
    __shared__ float Bs[GBK * GLD];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int64_t m0 = (int64_t)blockIdx.y * GBM, n0 = (int64_t)blockIdx.x * GBN;
    const int wm = (wid >> 1) * 64, wn = (wid & 1) * 64;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    const int64_t kbeg = (int64_t)blockIdx.z * kchunk;
    const int64_t kend = kbeg + kchunk < K ? kbeg + kchunk : K;
    for (int64_t k0 = kbeg; k0 < kend; k0 += GBK) {
        __syncthreads();
        generic_stage<T, AK>(A, lda, m0, e.M, k0, kend, As, tid);
        generic_stage<T, BKM>(B, ldb, n0, e.N, k0, kend, Bs, tid);
        __syncthreads();
#pragma unroll
        for (int s = 0; s < GBK / 4; ++s) {
            const int kk = 4 * s + (lane >> 4);
            float af[4], bfr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = As[kk * GLD + wm + 16 * i + (lane & 15)];
#pragma unroll
            for (int j = 0; j < 4; ++j) bfr[j] = Bs[kk * GLD + wn + 16 * j + (lane & 15)];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
    }
    if (gridDim.z > 1) {   // split-K: raw partial sums to this split's slab
        float* slab = e.slab + (int64_t)blockIdx.z * e.M * e.N;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t n = n0 + wn + 16 * j + (lane & 15);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t m = m0 + wm + 16 * i + 4 * (lane >> 4) + r;
                    if (m < e.M && n < e.N) slab[m * e.N + n] = acc[i][j][r];
                }
            }
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            epi_store4<T>(e, m0 + wm + 16 * i + 4 * (lane >> 4), n0 + wn + 16 * j + (lane & 15), acc[i][j]);
}

// Shapes where the 256 x 256 kernel is the faster one (tools/gemm_bench.py --tile both: +4..16 % on 4096^3 / 8192^3,
// slower whenever its grid leaves CUs idle or K is short): whole rounds of 256 workgroups, long K, no edge tiles.
static bool big_tile_pays(int64_t M, int64_t N, int64_t K, int64_t n_wg_big) {
    if (M % 256 || N % 256 || K < 2048 || n_wg_big < 256) return false;
    const int64_t rounds = (n_wg_big + 255) / 256;
    return rounds * 256 * 100 <= n_wg_big * 110;      // at most 10 % of the last round idle
}

// ---- the persistent 256-row-tile kernel (gemm_p8.hip) ----
struct P8Plan {
    bool use;
    int ntw, split, grid;
};
P8Plan vaw_p8_plan(int64_t M, int64_t N, int64_t K, bool plain_f32, bool want_colsum, int64_t ws_floats, int force);
void vaw_sm_launch(int mb, int nb, int stages, int b_kmajor, int64_t M, int64_t N, int64_t K, const bf16_t* a, int64_t lda,
                   const bf16_t* b, int64_t ldb, const EpiDev& e, hipStream_t s);      // gemm_sm.hip
bool vaw_p8_conv(int mode, const bf16_t* act, const bf16_t* act2, const bf16_t* w, void* out, int B, int H, int W, int Ci, int Co,
                 EpiDev e, float* workspace, int64_t workspace_floats, int force, hipStream_t s, float* bias_grad, float bias_beta,
                 int* bias_done);
void vaw_p8_launch(const P8Plan& pl, int a_kmajor, int b_kmajor, int64_t M, int64_t N, int64_t K, const bf16_t* a, int64_t lda,
                   const bf16_t* b, int64_t ldb, const EpiDev& e, hipStream_t s);

// the parked-drain kernel (gemm_pd.hip)
int vaw_pd_epi_kind(const EpiDev& e, bool a_kmajor, bool b_kmajor, int64_t M, int64_t N, int64_t K);
int vaw_pd_pick_ntw(int64_t M, int64_t N, int cus_avail);
void vaw_pd_launch(int ntw, int epi, int b_kmajor, int64_t M, int64_t N, int64_t K, const bf16_t* a, int64_t lda, const bf16_t* b,
                   int64_t ldb, const EpiDev& e, int cus_avail, hipStream_t s);
int vaw_p8_cus_available();
// the warp-specialised kernel (gemm_ws.hip)
void vaw_ws_launch(int ntw, int epi, int b_kmajor, int64_t M, int64_t N, int64_t K, const bf16_t* a, int64_t lda, const bf16_t* b,
                   int64_t ldb, const EpiDev& e, int cus, hipStream_t s);

static int g_force_generic = 0;
extern "C" void vaw_debug_force_generic_gemm(int on) { g_force_generic = on; }
// bf16 MFMA tile choice: -1 = by shape (default), 0 = always 128 x 128, 1 = always the 256 x 256 ring kernel,
// 2 / 3 = always the persistent kernel with 256 / 192 columns, 4 = always the persistent kernel (width by shape),
// 5-8 = the small-M ring kernel, 9 / 10 / 11 = the parked-drain kernel wherever it applies (width by shape / 256 / 192 columns).
// Env VAW_GEMM_BIG seeds it.
static int g_gemm_tile = -2;
extern "C" void vaw_debug_gemm_tile(int mode) { g_gemm_tile = mode; }

static bool takes_fast_path(vaw_dtype dt, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B,
                            int64_t ldb) {
    // any M and N (edge tiles are predicated), as long as rows are whole 16-byte chunks
    return dt == VAW_BF16 && !g_force_generic && N % 8 == 0 && K % BK == 0 && lda % 8 == 0 && ldb % 8 == 0 &&
           (((uintptr_t)A | (uintptr_t)B) & 15) == 0 && M >= 16 && N >= 16;
}
static bool fast_layout_ok(int a_kmajor, int64_t M) { return a_kmajor || M % 8 == 0; }
extern "C" int vaw_gemm_uses_bf16_mfma(vaw_dtype dt, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda,
                                       const void* B, int64_t ldb) {
    return takes_fast_path(dt, M, N, K, A, lda, B, ldb) ? 1 : 0;
}

// Split-K factor: only for plain f32-output epilogues (the weight gradients: long K = B*T, few output tiles),
// sized so the launch has ~2 workgroups per CU, each split keeping >= 256 of K, within the workspace.
static int pick_split(int64_t tiles, int64_t K, int64_t MN, int64_t ws_floats, bool plain_f32) {
    if (!plain_f32 || ws_floats <= 0) return 1;
    int64_t s = 512 / tiles;
    if (s > K / 256) s = K / 256;
    if (s > ws_floats / MN) s = ws_floats / MN;
    if (s > 64) s = 64;
    return s < 2 ? 1 : (int)s;
}

// rowsum_a_out without the fused path: A stored [K][M] (a_kmajor = 0) is a column sum over its K rows.
static int rowsum_a_separate(vaw_dtype dt, int a_kmajor, int64_t M, int64_t K, const void* A, int64_t lda, float* out, float beta,
                             float* workspace, int64_t workspace_floats, vaw_stream stream) {
    VAW_CHECK_ARG(!a_kmajor, "gemm: rowsum_a_out is defined for a_kmajor = 0 (weight-gradient layout) only");
    return vaw_colsum(dt, A, K, M, lda, out, beta, workspace, workspace_floats, stream);
}

// split-K launches whose split count divides 8 use the K-range-per-XCD mapping of gemm_bf16_kernel (xcd_parts = 8 / split)
static int xcd_parts_for(int split) {
    static int on = -1;
    if (on < 0) { const char* v = getenv("VAW_GEMM_XCDSPLIT"); on = v ? atoi(v) : 1; }
    return (on && (split == 2 || split == 4 || split == 8)) ? 8 / split : 0;
}

extern "C" int vaw_gemm(vaw_dtype dt, int a_kmajor, int b_kmajor, int64_t M, int64_t N, int64_t K, const void* A,
                        int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, const vaw_epilogue* ep,
                        float* workspace, int64_t workspace_floats, vaw_stream stream) {
    VAW_CHECK_ARG(M > 0 && N > 0 && K > 0 && A && B && C, "gemm: bad sizes M=%ld N=%ld K=%ld", (long)M, (long)N, (long)K);
    VAW_CHECK_ARG(M < (1LL << 31) && N < (1LL << 31), "gemm: M, N must fit 31 bits");
    VAW_CHECK_ARG(lda >= (a_kmajor ? K : M) && ldb >= (b_kmajor ? K : N) && ldc >= N, "gemm: leading dimension too small");
    EpiDev e{};
    e.alpha = 1.f;
    if (ep) {
        e.bias = ep->bias; e.act = ep->act; e.aux_in = ep->aux_in; e.aux_out = ep->aux_out; e.gate = ep->gate;
        e.gate_ld = ep->gate_ld; e.resid = ep->resid; e.rowadd = ep->rowadd; e.rpb = ep->rows_per_batch;
        e.alpha = ep->alpha; e.beta = ep->beta; e.out_f32 = ep->out_f32; e.resid_act = ep->resid_is_act;
    }
    float* const colsum_final = ep ? ep->colsum_out : nullptr;
    float* const colsum_part = ep ? ep->colsum_partial_out : nullptr;      // deferred fold: partial rows stay with the caller
    VAW_CHECK_ARG(!colsum_part || (!colsum_final && ep->colsum_rows_out), "gemm: colsum_partial_out excludes colsum_out and needs colsum_rows_out");
    const bool colsum_out = colsum_final || colsum_part;                  // "this launch carries column sums"
    float* const colsum_dst = colsum_part ? colsum_part : workspace;      // where the kernels leave their partial rows
    const float colsum_beta = ep ? ep->colsum_beta : 0.f;
    // colsum_rows_out is in / out: the caller states the capacity of colsum_partial_out in rows, the call answers with the rows
    // written -- checked BEFORE anything is launched (a kernel path that needs more rows than the buffer has is an error)
    const int64_t colsum_cap = colsum_part ? *ep->colsum_rows_out : 0;
#define CS_CAP_CHECK(R) VAW_CHECK_ARG(!colsum_part || (R) <= colsum_cap, "gemm: colsum_partial_out holds %ld rows, this launch writes %ld", (long)colsum_cap, (long)(R))
    auto fold_colsum = [&](int64_t R) -> int {
        if (colsum_part) { *ep->colsum_rows_out = R; return VAW_OK; }
        return vaw_reduce_rows(workspace, R, N, colsum_final, colsum_beta, stream);
    };
    float* rowsum_out = ep ? ep->rowsum_a_out : nullptr;
    const float rowsum_beta = ep ? ep->rowsum_a_beta : 0.f;
    VAW_CHECK_ARG(!rowsum_out || (workspace && workspace_floats >= 64 * M), "gemm: rowsum_a_out needs a workspace");
    VAW_CHECK_ARG(e.act >= 0 && e.act <= 2, "gemm: unknown act %d", e.act);
    VAW_CHECK_ARG(e.act != 2 || e.aux_in, "gemm: act=2 needs aux_in");
    VAW_CHECK_ARG(!(e.gate || e.rowadd) || e.rpb > 0, "gemm: gate/rowadd need rows_per_batch");
    VAW_CHECK_ARG(e.beta == 0.f || e.out_f32 || dt == VAW_F32, "gemm: beta needs f32 output");
    if (e.rpb <= 0) e.rpb = 1;
    if (dt == VAW_F32) e.out_f32 = 1;
    e.M = M; e.N = N; e.ldc = ldc; e.C = C; e.slab = workspace;
    {
        static int dbg = -1;
        if (dbg < 0) { const char* v = getenv("VAW_GEMM_DEBUG"); dbg = v ? atoi(v) : 0; }
        e.debug = dbg;
        static int depi = -1;
        if (depi < 0) { const char* v = getenv("VAW_GEMM_EPI"); depi = v ? atoi(v) : 1; }
        e.direct_epi = depi;
    }
    hipStream_t s = (hipStream_t)stream;
    const bool plain_f32 = e.out_f32 && !e.bias && !e.act && !e.aux_out && !e.gate && !e.resid && !e.rowadd && N % 4 == 0 &&
                           ldc % 4 == 0 && ((uintptr_t)C & 15) == 0;

    // the vector epilogue of the fast path needs every epilogue operand 16-byte aligned
    const bool epi_aligned = ldc % 8 == 0 && e.gate_ld % 4 == 0 &&
                             ((((uintptr_t)C | (uintptr_t)e.bias | (uintptr_t)e.aux_in | (uintptr_t)e.aux_out |
                                (uintptr_t)e.gate | (uintptr_t)e.resid | (uintptr_t)e.rowadd) & 15) == 0);
    VAW_CHECK_ARG(!colsum_final || (workspace && workspace_floats >= vaw_colsum_workspace_floats(M, N) &&
                                    workspace_floats >= ((M + 127) / 128) * N),
                  "gemm: colsum_out needs a workspace of max(ceil(M/128), ceil(M/512))*N floats");
    if (takes_fast_path(dt, M, N, K, A, lda, B, ldb) && epi_aligned && fast_layout_ok(a_kmajor, M)) {
        // small M (strong-scaling batches: a few thousand token rows): 64-row tiles with a deep LDS-DMA ring (gemm_sm.hip)
        // -- the launches the 128- and 256-row kernels can only give a quarter of the chip, one exposed memory latency per K step
        {
            static int sm_max_m = -1, sm_nb = 0, sm_st = 0, sm_wide_m = 0;
            if (sm_max_m < 0) {
                const char* v = getenv("VAW_SM_MAX_M"); sm_max_m = v ? atoi(v) : 8192;
                v = getenv("VAW_SM_WIDE_M"); sm_wide_m = v ? atoi(v) : 0;        // 128 x 128 tiles for the wide launches up to this M (0 = off)
                v = getenv("VAW_SM_NB"); sm_nb = v ? atoi(v) : 0;
                v = getenv("VAW_SM_STAGES"); sm_st = v ? atoi(v) : 0;
            }
            const int64_t rows64 = (M + 63) / 64;
            const bool cs_room = !colsum_out || colsum_part || workspace_floats >= rows64 * N;
            // measured (tools/gemm_bench.py --m 2048 / 4096 / 8192 --tile sm, DiT-B/4 shapes): 1.2-1.8x faster than the 128- and
            // 256-row kernels on the launches that give those less than one workgroup per CU (the 768-wide layers up to 4096
            // rows: 16.0 -> 11.3, 39.4 -> 24.4, 30.2 -> 17.2, 34.2 -> 21.1 us at 2048 rows), slower on the wide ones (its 64-row
            // tiles move twice the operand bytes per MFMA): taken only below one 128 x 128 tile per CU
            // (and only for the K of the blocks' Linear layers: the long-K launches -- adaLN's input gradient, K = 6 L D -- keep
            //  their split-K path: 161 us on 48 workgroups here against 40 + 38 us split)
            // With the LDS-staged epilogue (round 3, late) the ring kernel wins on every 768-wide layer up to 4096 rows, and up to
            // 8192 rows on those with K = 768 (proj: 33.2 -> 24.4, 23.3 -> 17.9 us) and on fc2's forward (K = 3072: 72.7 -> 61.2 with
            // 128-column tiles); the wide layers (N >= 2304) tie at 2048 rows and lose above: they keep the other kernels
            const bool sm_few_tiles = N <= 1024 && K <= 4096 &&
                                      (M <= 4096 || (M <= 8192 && (K <= 1024 || b_kmajor)));
            // wide layers at small M (fc1, fc2's GELU' input gradient, qkv at 2048-4096 rows: 128-288 items of the larger kernels):
            // 128 x 128 tiles on the same ring
            const bool sm_wide = !sm_few_tiles && M <= sm_wide_m && K <= 4096 && ((M + 127) / 128) * ((N + 127) / 128) <= 1024;
            const bool sm_forced = g_gemm_tile >= 5 && g_gemm_tile <= 8;     // vaw_debug_gemm_tile: 5 always, 6 / 7 / 8 always with 64 x 64 / 64 x 128 / 128 x 128 tiles
            if (a_kmajor && ((M <= sm_max_m && (sm_few_tiles || sm_wide) && g_gemm_tile < 0) || sm_forced) && !rowsum_out && cs_room && N % 8 == 0) {
                // 64 x 128 tiles when they still give every CU a workgroup, 64 x 64 otherwise; ring depth by the LDS it leaves:
                // 3 stages = two (64 x 128) or three (64 x 64) workgroups per CU for multi-round launches, 4 for single rounds
                const int mb = g_gemm_tile == 8 ? 2 : (g_gemm_tile >= 5 ? 1 : (sm_wide && !sm_few_tiles) ? 2 : 1);
                const int nb = mb == 2 ? 2 : g_gemm_tile == 6 ? 1 : g_gemm_tile == 7 ? 2 : sm_nb ? sm_nb : ((M >= 4096 && K >= 2048) ? 2 : 1);
                const int64_t rows_t = (M + 64 * mb - 1) / (64 * mb);
                const int64_t tiles = rows_t * ((N + 64 * nb - 1) / (64 * nb));
                const int stages = sm_st ? sm_st : (tiles > 256 ? 3 : 4);
                EpiDev es = e;
                CS_CAP_CHECK(rows_t);
                if (colsum_out) es.colpart = colsum_dst;
                vaw_sm_launch(mb, nb, stages, b_kmajor, M, N, K, (const bf16_t*)A, lda, (const bf16_t*)B, ldb, es, s);
                VAW_CHECK_LAUNCH("gemm_sm");
                if (colsum_out) return fold_colsum(rows_t);
                return VAW_OK;
            }
        }
        const int tiles_n = (int)((N + BN - 1) / BN);
        const int64_t n_wg = ((M + BM - 1) / BM) * tiles_n;
        VAW_CHECK_ARG(n_wg < (1LL << 31), "gemm: grid too large");
        // stage depth: 64 for long K (weight gradients), 32 for the K <= 1024 forward / input-gradient launches
        static int bk_env = -1;
        if (bk_env < 0) { const char* v = getenv("VAW_GEMM_BK"); bk_env = v ? atoi(v) : 0; }
        // measured on MI355X (tools/gemm_bench.py, DiT-B/4 shapes): the 32-deep stage (3 workgroups per CU) wins for
        // the input-gradient layout (k-major x mn-major) when there are at least ~3 tiles per CU to overlap its K steps
        // (768 output tiles at batch 256); with fewer tiles (per-GPU batches of 32 / 64 under strong scaling: 96 / 192 tiles)
        // every K step is exposed latency and the 64-deep stage halves their number (step 7.05 -> 6.46 ms at batch 32)
        const bool many_tiles = n_wg >= 3 * 256;
        const int bkt = bk_env == 32 || bk_env == 64 ? bk_env : ((a_kmajor && !b_kmajor && K <= 4096 && many_tiles) ? 32 : 64);
        const int nk_total = (int)(K / bkt);
        const bool fused_rowsum = rowsum_out && !a_kmajor && !colsum_out;     // row sums of A ride on the MFMA kernel
        int split = colsum_out ? 1 : pick_split(n_wg, K, M * N, workspace_floats - (fused_rowsum ? 64 * M : 0), plain_f32);
        // small-M input gradients (per-GPU batches of 32 / 64 under strong scaling: 96-192 tiles, K up to 3072): with one
        // workgroup per CU every K step is exposed DMA latency, so a plain bf16 result is K-split as well (f32 slabs, fixed-order
        // reduce that writes bf16) when K is long enough to pay for the slab round trip
        const bool plain_bf16 = !e.out_f32 && !e.bias && !e.act && !e.aux_out && !e.gate && !e.resid && !e.rowadd && e.beta == 0.f &&
                                N % 4 == 0 && ldc % 4 == 0 && ((uintptr_t)C & 7) == 0;
        if (split == 1 && plain_bf16 && !colsum_out && !rowsum_out && workspace && K >= 2048 && n_wg <= 256) {
            int64_t sp = 512 / n_wg;
            if (sp > K / 512) sp = K / 512;
            if (sp > workspace_floats / (M * N)) sp = workspace_floats / (M * N);
            if (sp > 8) sp = 8;
            if (sp >= 2) split = (int)sp;
        }
        if (colsum_out) e.colpart = colsum_dst;
        if (split > 1) {   // no empty splits
            const int per = (nk_total + split - 1) / split;
            split = (nk_total + per - 1) / per;
        }
        float* rowpart = nullptr;
        if (fused_rowsum) rowpart = e.rowpart = workspace + (split > 1 ? (int64_t)split * M * N : 0);
        const bf16_t* a = (const bf16_t*)A;
        const bf16_t* b = (const bf16_t*)B;
        const int xparts = xcd_parts_for(split);
        dim3 grid((unsigned)n_wg, (unsigned)split);
        if (xparts) grid = dim3((unsigned)(8 * ((n_wg + xparts - 1) / xparts)), 1);
#define LAUNCH_FAST(AKv, BKv, BKTv)                                                                                   \
    do {                                                                                                              \
        static bool attr_done = false;                                                                                \
        const int lds = FastCfg<BKTv>::lds_bytes;                                                                     \
        if (!attr_done) {                                                                                             \
            (void)hipFuncSetAttribute((const void*)gemm_bf16_kernel<AKv, BKv, BKTv, 0>,                               \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds);                               \
            attr_done = true;                                                                                         \
        }                                                                                                             \
        gemm_bf16_kernel<AKv, BKv, BKTv, 0><<<grid, 256, lds, s>>>(a, lda, b, ldb, nk_total, tiles_n, (int)n_wg, split, e, \
                                                                    ConvGeom{}, xparts);                             \
    } while (0)
#define LAUNCH_FAST_BK(BKTv)                                         \
    do {                                                             \
        if (a_kmajor && b_kmajor) LAUNCH_FAST(true, true, BKTv);     \
        else if (a_kmajor && !b_kmajor) LAUNCH_FAST(true, false, BKTv); \
        else if (!a_kmajor && b_kmajor) LAUNCH_FAST(false, true, BKTv); \
        else LAUNCH_FAST(false, false, BKTv);                        \
    } while (0)
        if (g_gemm_tile == -2) { const char* v = getenv("VAW_GEMM_BIG"); g_gemm_tile = v ? atoi(v) : -1; }
        {
            // parked-drain kernel (gemm_pd_kernel.h): the un-split forward / input-gradient launches of the Linear layers whose
            // epilogue it can hide under the next tile's K loop.  Measured level with or behind the 256-row
            // kernel on every DiT-B/4 shape, so the automatic choice is OFF: VAW_GEMM_PD=1 switches it on by shape,
            // vaw_debug_gemm_tile(9 / 10 / 11) forces it (tests, A/B runs).
            static int pd_auto = -1;
            if (pd_auto < 0) { const char* v = getenv("VAW_GEMM_PD"); pd_auto = v ? atoi(v) : 0; }
            const bool ws_forced = g_gemm_tile >= 12 && g_gemm_tile <= 14;       // 12 / 13 / 14: the warp-specialised kernel (width by shape / 256 / 192)
            static int ws_auto = -1;
            if (ws_auto < 0) { const char* v = getenv("VAW_GEMM_WS"); ws_auto = v ? atoi(v) : 0; }
            const bool pd_forced = (g_gemm_tile >= 9 && g_gemm_tile <= 11) || ws_forced;
            const int64_t rows64 = (M + 63) / 64;
            EpiDev epd = e;          // (e.colpart is set above when this launch carries column sums)
            const int pd_kind = (pd_forced || ((pd_auto || ws_auto) && g_gemm_tile == -1)) && bk_env == 0 && !rowsum_out &&
                                        (!colsum_out || colsum_part || workspace_floats >= rows64 * N)
                                    ? vaw_pd_epi_kind(epd, a_kmajor != 0, b_kmajor != 0, M, N, K) : -1;
            if (pd_kind >= 0) {
                const int cus = vaw_p8_cus_available();
                const int ntw = (g_gemm_tile == 10 || g_gemm_tile == 13) ? 4 : (g_gemm_tile == 11 || g_gemm_tile == 14) ? 3 : vaw_pd_pick_ntw(M, N, cus);
                const int64_t items = ((M + 127) / 128) * ((N + 64 * ntw - 1) / (64 * ntw));
                // by shape: at least two rounds of workgroups (the first tile of a workgroup has nothing to hide its epilogue
                // under... the last one's leaves in the open), K of the blocks' Linear layers
                const bool pd_shape = items >= 2 * cus && K >= 768 && K <= 4096;
                if (pd_forced || pd_shape) {
                    CS_CAP_CHECK(rows64);
                    static int nt_aux = -1;
                    if (nt_aux < 0) { const char* v = getenv("VAW_P8_NT_AUX"); nt_aux = (v && atoi(v) == 0) ? 0 : 1; }
                    epd.nt_off = 1;
                    epd.nt_aux = nt_aux;
                    epd.colpart = colsum_out ? colsum_dst : nullptr;
                    if (ws_forced || (ws_auto && !pd_forced)) vaw_ws_launch(ntw, pd_kind, b_kmajor, M, N, K, a, lda, b, ldb, epd, cus, s);
                    else vaw_pd_launch(ntw, pd_kind, b_kmajor, M, N, K, a, lda, b, ldb, epd, cus, s);
                    VAW_CHECK_LAUNCH("gemm_pd");
                    if (colsum_out) return fold_colsum(rows64);
                    return VAW_OK;
                }
            }
        }
        {
            const int force = g_gemm_tile == -1 ? -1 : g_gemm_tile == 4 ? 1 : (g_gemm_tile == 2 || g_gemm_tile == 3) ? g_gemm_tile : 0;
            const bool p8_epi_ok = !(e.act == 2 && e.gate) && !(e.resid && e.rowadd);     // gemm_epi.h: EpiOps has two slots
            const P8Plan pl = (bk_env == 0 && !fused_rowsum && p8_epi_ok)
                                  ? vaw_p8_plan(M, N, K, plain_f32 || (plain_bf16 && !rowsum_out && K >= 2048 && workspace != nullptr),
                                                colsum_out, workspace_floats, force)
                                  : P8Plan{false, 4, 1, 0};
            // small M (strong-scaling batches): a persistent launch that gives only half the CUs an item loses to the 128 x 128
            // kernel's 2-3 workgroups per CU (fc1 forward at 2048 rows: 28.7 vs 22.0 us, fc2's GELU' input gradient 29.7 vs 26.5)
            const bool p8_half_empty = force < 0 && M <= 4096 && pl.use && pl.grid < 200 &&
                                       ((M + 255) / 256) * ((N + 64 * pl.ntw - 1) / (64 * pl.ntw)) * pl.split < 200;
            if (pl.use && !p8_half_empty) {
                CS_CAP_CHECK((M + 127) / 128);
                EpiDev ep8 = e;
                static int nt_off = -1;
                // default: epilogue stores with the default cache policy (measured: nt costs 5-15 % on the f32 gated-residual and
                // the 192-column launches, whose row segments are not whole 128-byte lines, and the next kernel re-reads the
                // output from L2 / MALL anyway); VAW_P8_NT=1 switches the non-temporal hint on
                if (nt_off < 0) { const char* v = getenv("VAW_P8_NT"); nt_off = (v && atoi(v) == 1) ? 0 : 1; }
                ep8.nt_off = nt_off;
                static int nt_aux = -1;
                if (nt_aux < 0) { const char* v = getenv("VAW_P8_NT_AUX"); nt_aux = (v && atoi(v) == 0) ? 0 : 1; }      // default on: -0.5 % on the DiT-B/4 step (13.89 -> 13.83, 13.95 -> 13.88 ms, one box)
                ep8.nt_aux = nt_aux;
                ep8.colpart = colsum_out ? colsum_dst : nullptr;
                ep8.rowpart = nullptr;
                vaw_p8_launch(pl, a_kmajor, b_kmajor, M, N, K, a, lda, b, ldb, ep8, s);
                if (pl.split > 1 && !e.out_f32)      // plain bf16 result (input gradients of half-full launches): slabs -> bf16
                    splitk_reduce_kernel<bf16_t><<<ceil_div(M * N / 4, 256) > 2048 ? 2048 : ceil_div(M * N / 4, 256), 256, 0, s>>>(
                        workspace, pl.split, M, N, ldc, C, e.alpha, 0.f, 0);
                else if (pl.split > 1)
                    splitk_reduce_kernel<float><<<ceil_div(M * N / 4, 256) > 2048 ? 2048 : ceil_div(M * N / 4, 256), 256, 0, s>>>(
                        workspace, pl.split, M, N, ldc, C, e.alpha, e.beta, 1);
                VAW_CHECK_LAUNCH("gemm_p8");
                if (rowsum_out) {
                    const int rc = rowsum_a_separate(dt, a_kmajor, M, K, A, lda, rowsum_out, rowsum_beta, workspace, workspace_floats, stream);
                    if (rc != VAW_OK) return rc;
                }
                if (colsum_out) return fold_colsum((M + 127) / 128);
                return VAW_OK;
            }
        }
        const int64_t tiles_nb = (N + BIG_BN - 1) / BIG_BN, n_wgb = ((M + BIG_BM - 1) / BIG_BM) * tiles_nb;
        const bool use_big = bk_env == 0 && !fused_rowsum && (g_gemm_tile == 1 || (g_gemm_tile == -1 && big_tile_pays(M, N, K, n_wgb)));
        if (use_big) {
            CS_CAP_CHECK((M + BIG_BM - 1) / BIG_BM);
            const int nkb = (int)(K / 32);
            int splitb = colsum_out ? 1 : pick_split(n_wgb * 2, K, M * N, workspace_floats, plain_f32);
            if (splitb > 1) {
                const int per = (nkb + splitb - 1) / splitb;
                splitb = (nkb + per - 1) / per;
            }
            dim3 gridb((unsigned)n_wgb, (unsigned)splitb);
#define LAUNCH_BIG(AKv, BKv)                                                                                             \
    do {                                                                                                                 \
        static bool attr_done = false;                                                                                   \
        if (!attr_done) {                                                                                                \
            (void)hipFuncSetAttribute((const void*)gemm_bf16_big_kernel<AKv, BKv>,                                       \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, BIG_LDS);                              \
            attr_done = true;                                                                                            \
        }                                                                                                                \
        gemm_bf16_big_kernel<AKv, BKv><<<gridb, 512, BIG_LDS, s>>>(a, lda, b, ldb, nkb, (int)tiles_nb, (int)n_wgb, splitb, e); \
    } while (0)
            if (a_kmajor && b_kmajor) LAUNCH_BIG(true, true);
            else if (a_kmajor && !b_kmajor) LAUNCH_BIG(true, false);
            else if (!a_kmajor && b_kmajor) LAUNCH_BIG(false, true);
            else LAUNCH_BIG(false, false);
            if (splitb > 1)
                splitk_reduce_kernel<float><<<ceil_div(M * N / 4, 256) > 2048 ? 2048 : ceil_div(M * N / 4, 256), 256, 0, s>>>(
                    workspace, splitb, M, N, ldc, C, e.alpha, e.beta, 1);
            VAW_CHECK_LAUNCH("gemm_bf16_big");
            if (rowsum_out) {
                const int rc = rowsum_a_separate(dt, a_kmajor, M, K, A, lda, rowsum_out, rowsum_beta, workspace, workspace_floats, stream);
                if (rc != VAW_OK) return rc;
            }
            if (colsum_out) return fold_colsum((M + BIG_BM - 1) / BIG_BM);
            return VAW_OK;
        }
        CS_CAP_CHECK((M + BM - 1) / BM);
        if (bkt == 32) LAUNCH_FAST_BK(32);
        else LAUNCH_FAST_BK(64);
        if (split > 1 && !e.out_f32)
            splitk_reduce_kernel<bf16_t><<<ceil_div(M * N / 4, 256) > 2048 ? 2048 : ceil_div(M * N / 4, 256), 256, 0, s>>>(
                workspace, split, M, N, ldc, C, e.alpha, 0.f, 0);
        else if (split > 1)
            splitk_reduce_kernel<float><<<ceil_div(M * N / 4, 256) > 2048 ? 2048 : ceil_div(M * N / 4, 256), 256, 0, s>>>(
                workspace, split, M, N, ldc, C, e.alpha, e.beta, 1, rowpart, rowsum_out, rowsum_beta);
        VAW_CHECK_LAUNCH("gemm_bf16");
        if (fused_rowsum && split == 1) {
            const int rc = vaw_reduce_rows(rowpart, split, M, rowsum_out, rowsum_beta, stream);
            if (rc != VAW_OK) return rc;
        } else if (rowsum_out && !fused_rowsum) {
            const int rc = rowsum_a_separate(dt, a_kmajor, M, K, A, lda, rowsum_out, rowsum_beta, workspace, workspace_floats, stream);
            if (rc != VAW_OK) return rc;
        }
        if (colsum_out) return fold_colsum((M + BM - 1) / BM);
        return VAW_OK;
    }
    const int64_t tiles = (int64_t)ceil_div(N, GBN) * ceil_div(M, GBM);
    int split = colsum_out ? 1 : pick_split(tiles, K, M * N, workspace_floats, plain_f32);
    int64_t kchunk = K;
    if (split > 1) {
        kchunk = ((K + split - 1) / split + GBK - 1) / GBK * GBK;
        split = (int)((K + kchunk - 1) / kchunk);
    }
    CS_CAP_CHECK(1);
    dim3 grid(ceil_div(N, GBN), ceil_div(M, GBM), split);
#define LAUNCH_GEN(T, AKv, BKv) \
    gemm_generic_kernel<T, AKv, BKv><<<grid, 256, 0, s>>>((const T*)A, lda, (const T*)B, ldb, K, kchunk, e)
#define LAUNCH_GEN_T(T)                                          \
    do {                                                         \
        if (a_kmajor && b_kmajor) LAUNCH_GEN(T, true, true);     \
        else if (a_kmajor && !b_kmajor) LAUNCH_GEN(T, true, false); \
        else if (!a_kmajor && b_kmajor) LAUNCH_GEN(T, false, true); \
        else LAUNCH_GEN(T, false, false);                        \
    } while (0)
    if (dt == VAW_F32) LAUNCH_GEN_T(float);
    else LAUNCH_GEN_T(bf16_t);
    if (split > 1)
        splitk_reduce_kernel<float><<<ceil_div(M * N / 4, 256) > 2048 ? 2048 : ceil_div(M * N / 4, 256), 256, 0, s>>>(
            workspace, split, M, N, ldc, C, e.alpha, e.beta, 1);
    VAW_CHECK_LAUNCH("gemm_generic");
    if (rowsum_out) {
        const int rc = rowsum_a_separate(dt, a_kmajor, M, K, A, lda, rowsum_out, rowsum_beta, workspace, workspace_floats, stream);
        if (rc != VAW_OK) return rc;
    }
    if (colsum_part) {   // generic path, deferred fold: one complete row of column sums as the only "partial" row
        VAW_CHECK_ARG(workspace && workspace_floats >= vaw_colsum_workspace_floats(M, N), "gemm: colsum_partial_out on the generic path needs the workspace");
        *ep->colsum_rows_out = 1;
        return vaw_colsum(e.out_f32 ? VAW_F32 : dt, C, M, N, ldc, colsum_part, 0.f, workspace, workspace_floats, stream);
    }
    if (colsum_out)   // generic path: a separate pass over the output just written
        return vaw_colsum(e.out_f32 ? VAW_F32 : dt, C, M, N, ldc, colsum_final, colsum_beta, workspace, workspace_floats, stream);
    return VAW_OK;
}


// =============================================================================================
// conv3x3 (stride 1, pad 1, NHWC) as implicit GEMM on the MFMA kernel.  Returns VAW_ERR_UNSUPPORTED when the
// shape needs the explicit path (f32 parity mode, channel counts that are not multiples of 64, ...).
//   mode 0  y[M,Co]   = conv(x; W) (+ epilogue)          act = x  [M,Ci]
//   mode 1  dx[M,Ci]  = conv^T(dy; W)                    act = dy [M,Co]
//   mode 2  dW[Co,9Ci] (f32) = beta*dW + dy^T . patches(x)   act = dy, act2 = x
// W is stored [Co][3][3][Ci] (channels-last), act dtype.
// =============================================================================================
extern "C" int vaw_conv3x3(vaw_dtype dt, int mode, const void* act, const void* act2, const void* w, void* out, int B, int H,
                           int W, int Ci, int Co, const vaw_epilogue* ep, float* workspace, int64_t workspace_floats,
                           vaw_stream stream) {
    VAW_CHECK_ARG(mode >= 0 && mode <= 2 && act && (w || mode == 2) && out && B > 0 && H > 0 && W > 0 && Ci > 0 && Co > 0,
                  "conv3x3: bad arguments");
    if (dt != VAW_BF16 || g_force_generic) return VAW_ERR_UNSUPPORTED;
    const int64_t Mpix = (int64_t)B * H * W;
    int64_t M, N, K;
    if (mode == 0) { M = Mpix; N = Co; K = 9LL * Ci; if (Ci % 64) return VAW_ERR_UNSUPPORTED; }
    else if (mode == 1) { M = Mpix; N = Ci; K = 9LL * Co; if (Co % 64) return VAW_ERR_UNSUPPORTED; }
    else { M = Co; N = 9LL * Ci; K = Mpix; if (Ci % 8 || Mpix % 64 || Co % 8 || !act2) return VAW_ERR_UNSUPPORTED; }
    if (N % 8 || M < 16 || N < 16 || Mpix >= (1LL << 31)) return VAW_ERR_UNSUPPORTED;
    if ((((uintptr_t)act | (uintptr_t)act2 | (uintptr_t)w | (uintptr_t)out) & 15) != 0) return VAW_ERR_UNSUPPORTED;
    EpiDev e{};
    e.alpha = 1.f;
    if (ep) {
        e.bias = ep->bias; e.act = ep->act; e.aux_in = ep->aux_in; e.aux_out = ep->aux_out; e.gate = ep->gate;
        e.gate_ld = ep->gate_ld; e.resid = ep->resid; e.rowadd = ep->rowadd; e.rpb = ep->rows_per_batch;
        e.alpha = ep->alpha; e.beta = ep->beta; e.out_f32 = ep->out_f32; e.resid_act = ep->resid_is_act;
    }
    if (e.rpb <= 0) e.rpb = 1;
    {
        static int depi = -1;
        if (depi < 0) { const char* v = getenv("VAW_GEMM_EPI"); depi = v ? atoi(v) : 1; }
        e.direct_epi = depi;
    }
    if (mode == 2) e.out_f32 = 1;
    VAW_CHECK_ARG(e.beta == 0.f || e.out_f32, "conv3x3: beta needs f32 output");
    float* colsum_out = ep ? ep->colsum_out : nullptr;
    const float colsum_beta = ep ? ep->colsum_beta : 0.f;
    const int64_t ldc = N;
    e.M = M; e.N = N; e.ldc = ldc; e.C = out; e.slab = workspace;
    const bool epi_aligned = e.gate_ld % 4 == 0 && ((((uintptr_t)e.bias | (uintptr_t)e.aux_in | (uintptr_t)e.aux_out |
                                                     (uintptr_t)e.gate | (uintptr_t)e.resid | (uintptr_t)e.rowadd) & 15) == 0);
    if (!epi_aligned) return VAW_ERR_UNSUPPORTED;
    VAW_CHECK_ARG(!colsum_out || (workspace && workspace_floats >= ((M + 127) / 128) * N), "conv3x3: colsum_out needs a workspace");
    hipStream_t s = (hipStream_t)stream;
    {   // the persistent 256-row-tile kernel first (gemm_p8_conv.hip); shapes it declines stay on the 128 x 128 kernel below
        if (g_gemm_tile == -2) { const char* v = getenv("VAW_GEMM_BIG"); g_gemm_tile = v ? atoi(v) : -1; }
        const int force = g_gemm_tile == -1 ? -1 : g_gemm_tile == 4 ? 1 : (g_gemm_tile == 2 || g_gemm_tile == 3) ? g_gemm_tile : 0;
        float* rowsum_out8 = ep ? ep->rowsum_a_out : nullptr;
        int bias_done = 0;
        if (!colsum_out && (mode != 2 || act2) && force != 0 &&
            vaw_p8_conv(mode, (const bf16_t*)act, (const bf16_t*)act2, (const bf16_t*)w, out, B, H, W, Ci, Co, e, workspace, workspace_floats,
                        force, s, mode == 2 ? rowsum_out8 : nullptr, ep ? ep->rowsum_a_beta : 0.f, &bias_done)) {
            VAW_CHECK_LAUNCH("conv3x3_p8");
            if (rowsum_out8 && !bias_done)     // bias gradient = column sums of dy [Mpix][Co], as a pass of its own
                return vaw_colsum(dt, act, Mpix, Co, Co, rowsum_out8, ep->rowsum_a_beta, workspace, workspace_floats, stream);
            return VAW_OK;
        }
    }
    const int tiles_n = (int)((N + BN - 1) / BN);
    const int64_t n_wg = ((M + BM - 1) / BM) * tiles_n;
    const int nk_total = (int)(K / 64);
    const bool plain_f32 = mode == 2 && !e.bias && !e.act;
    float* rowsum_out = ep ? ep->rowsum_a_out : nullptr;
    const float rowsum_beta = ep ? ep->rowsum_a_beta : 0.f;
    VAW_CHECK_ARG(!rowsum_out || mode == 2, "conv3x3: rowsum_a_out (bias gradient) belongs to mode 2");
    VAW_CHECK_ARG(!(colsum_out && mode == 2), "conv3x3: mode 2 takes rowsum_a_out, not colsum_out");
    const bool bias_grad = rowsum_out != nullptr;        // sum over pixels of dy (row sums of A = dy^T)
    int split = (colsum_out && !bias_grad) ? 1 : pick_split(n_wg, K, M * N, workspace_floats - (bias_grad ? 64 * M : 0), plain_f32);
    if (colsum_out && !bias_grad) e.colpart = workspace;
    if (split > 1) {
        const int per = (nk_total + split - 1) / split;
        split = (nk_total + per - 1) / per;
    }
    float* rowpart = nullptr;
    if (bias_grad) {
        VAW_CHECK_ARG(workspace && workspace_floats >= (split > 1 ? split * M * N : 0) + split * M, "conv3x3: bias gradient needs a workspace");
        rowpart = e.rowpart = workspace + (split > 1 ? split * M * N : 0);
    }
    const int xparts = xcd_parts_for(split);
    dim3 grid((unsigned)n_wg, (unsigned)split);
    if (xparts) grid = dim3((unsigned)(8 * ((n_wg + xparts - 1) / xparts)), 1);
    const ConvGeom cg{H, W, Ci, Co};
    const int lds = FastCfg<64>::lds_bytes;
#define LAUNCH_CONV(AKv, BKv, CV, Aptr, LDA, Bptr, LDB)                                                                  \
    do {                                                                                                                 \
        static bool attr_done = false;                                                                                   \
        if (!attr_done) {                                                                                                \
            (void)hipFuncSetAttribute((const void*)gemm_bf16_kernel<AKv, BKv, 64, CV>,                                   \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds);                                  \
            attr_done = true;                                                                                            \
        }                                                                                                                \
        gemm_bf16_kernel<AKv, BKv, 64, CV><<<grid, 256, lds, s>>>((const bf16_t*)(Aptr), LDA, (const bf16_t*)(Bptr), LDB, \
                                                                  nk_total, tiles_n, (int)n_wg, split, e, cg, xparts);   \
    } while (0)
    if (mode == 0) LAUNCH_CONV(true, true, 1, act, (int64_t)Ci, w, 9LL * Ci);
    else if (mode == 1) LAUNCH_CONV(true, false, 2, act, (int64_t)Co, w, 9LL * Ci);
    else LAUNCH_CONV(false, false, 3, act, (int64_t)Co, act2, (int64_t)Ci);
    if (split > 1)
        splitk_reduce_kernel<float><<<ceil_div(M * N / 4, 256) > 2048 ? 2048 : ceil_div(M * N / 4, 256), 256, 0, s>>>(
            workspace, split, M, N, ldc, out, e.alpha, e.beta, 1, rowpart, rowsum_out, rowsum_beta);
    VAW_CHECK_LAUNCH("conv3x3");
    if (bias_grad && split == 1) return vaw_reduce_rows(rowpart, split, M, rowsum_out, rowsum_beta, stream);
    if (colsum_out) return vaw_reduce_rows(workspace, (M + BM - 1) / BM, N, colsum_out, colsum_beta, stream);
    return VAW_OK;
}
