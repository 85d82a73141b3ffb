// Small-M bf16 MFMA GEMM: the Linear launches of a DiT step when the per-GPU batch is small (strong scaling of a global batch over
// 8 GPUs, reference main.py:166-180: per-GPU batch = batch_size // world_size; DiT-B/4 at 32-64 images = 2048-4096 token rows).
//
// Why a third kernel.  At M = 2048 a 768-wide layer has 8 x 4 tiles of the persistent kernel (gemm_p8_kernel.h) and 16 x 6 of
// the 128 x 128 one (gemm.hip): a quarter of the chip at best, each workgroup walking its K loop alone on its CU with ONE K tile
// of LDS-DMA in flight -- every K step costs a whole memory latency (measured: 28 us for 2048 x 768 x 768, 2.4 GFLOP).  Here:
//   * 64-row tiles, 64 or 128 columns wide: 384 / 192 workgroups for that launch, 256 threads = 4 waves as 2 (M) x 2 (N);
//   * a ring of STAGES (3 or 4) LDS stages of one 64-deep K tile each, filled by `buffer_load_dwordx4 ... lds`, with a COUNTED
//     s_waitcnt vmcnt(N) and a raw s_barrier per K tile: STAGES - 1 K tiles stay in flight, so the loop streams at the DMA
//     rate of the CU instead of paying a latency per step (cdna_hip_programming.md "Pipelining across barriers");
//   * LDS images, swizzles and fragment reads are those of the persistent kernel (8 KiB parts of 64 rows x 128 B, p8_frag);
//   * LDS-staged epilogue (accumulators -> f32 image over the dead ring -> 8 consecutive columns per thread, 16-byte accesses) with
//     the row epilogue of gemm_epi.h, column sums of the output (the next bias gradient) folded per tile row in a fixed order.
// Layouts: A k-major ([M][K]); B k-major ([N][K], forward) or mn-major ([K][N], input gradients).  Weight gradients keep the
// grouped persistent launch (K = tokens is long there and all layers together fill the chip).
#include "gemm_p8_kernel.h"

template <int NB, int MB = 1> struct SmCfg {
    static constexpr int BM = 64 * MB;
    static constexpr int BN = 64 * NB;
    static constexpr int stage_bytes = (MB + NB) * P8_PART;    // MB A parts (64 rows each) + NB B parts
    static constexpr int pieces = 2 * (MB + NB);               // DMA instructions per wave and K tile
};

// MB = 2 (128-row tiles, wave tile 64 x 32 NB): the wide launches of a small batch (fc1 / fc2's GELU' at 2048-4096 rows: 128 items
// of the persistent kernel, half the chip) as 128 x 128 tiles with the same deep ring.
template <bool BKM, int NB, int STAGES, int MB = 1>
__global__ void __launch_bounds__(256)
gemm_sm_kernel(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ B, int64_t ldb, int nk, int tiles_m, int tiles_n,
               EpiDev e) {
    using Cfg = SmCfg<NB, MB>;
    constexpr int NT = 2 * NB;                                  // 16-column MFMA tiles per wave (wave tile 32 MB x 32 NB)
    constexpr int MT = 2 * MB;                                  // 16-row MFMA tiles per wave
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [STAGES][A part | B parts]
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wid >> 1, wc = wid & 1;
    // XCD x (workgroups b = x mod 8) takes a contiguous run of tiles, column tile fastest: its workgroups share A row panels
    // (and the small weight operand) in that XCD's L2
    int tile;
    {
        const int n = tiles_m * tiles_n, b = blockIdx.x, x = b & 7, q = n >> 3, r = n & 7;
        tile = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
    }
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int64_t m0 = (int64_t)tm * Cfg::BM, n0 = (int64_t)tn * Cfg::BN;
    const int mvalid = e.M - m0 < Cfg::BM ? (int)(e.M - m0) : Cfg::BM;
    const int nvalid = e.N - n0 < Cfg::BN ? (int)(e.N - n0) : Cfg::BN;

    // ---- DMA stream: per-lane byte offsets fixed for the tile, a scalar offset walks K ----
    const __amdgpu_buffer_rsrc_t rs_a = epi_rsrc(A + m0 * lda);
    const __amdgpu_buffer_rsrc_t rs_b = epi_rsrc(BKM ? B + n0 * ldb : B + n0);
    unsigned off_a[MB][2], off_b[NB][2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int p = 0; p < MB; ++p) off_a[p][h] = p8_src_off<true>(false, p, wid + 4 * h, lane, lda, mvalid);
#pragma unroll
        for (int p = 0; p < NB; ++p) off_b[p][h] = p8_src_off<BKM>(false, p, wid + 4 * h, lane, ldb, nvalid);
    }
    const unsigned step_a = 128u, step_b = BKM ? 128u : (unsigned)(64 * ldb * 2);
    unsigned so_a = 0, so_b = 0;
    // (r4) no end state: past the last K tile the stream keeps issuing pieces at out-of-range offsets (zeros into a stage nobody reads
    // any more), so the K loop has ONE counted wait and no `if (iss < nk)` -- scalar branches in a K loop cost far more than they look
    // (gemm_pd_kernel.h); stage indices are carried, not computed modulo STAGES
    int iss = 0, iss_stage = 0;                                 // next K tile to issue, and its stage
    auto issue = [&]() __attribute__((always_inline)) {
        char* dst = smem + iss_stage * Cfg::stage_bytes;
        const bool live = iss < nk;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int p = 0; p < MB; ++p) p8_dma16(rs_a, dst + p * P8_PART + (wid + 4 * h) * 1024, live ? off_a[p][h] : EPI_OOB, so_a);
#pragma unroll
            for (int p = 0; p < NB; ++p) p8_dma16(rs_b, dst + (MB + p) * P8_PART + (wid + 4 * h) * 1024, live ? off_b[p][h] : EPI_OOB, so_b);
        }
        so_a += step_a;
        so_b += step_b;
        ++iss;
        iss_stage = iss_stage == STAGES - 1 ? 0 : iss_stage + 1;
    };
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s) issue();

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0, 0, 0, 0};

    int cstage = 0;
    for (int kt = 0; kt < nk; ++kt) {
        // K tile kt has landed once at most the (STAGES - 2) tiles issued after it are outstanding
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * Cfg::pieces) : "memory");
        __builtin_amdgcn_s_barrier();          // every wave's pieces of tile kt are in; everyone has finished reading tile kt - 1
        issue();                               // ... so its stage takes tile kt + STAGES - 1 (or out-of-range pieces past the end)
        const char* st = smem + cstage * Cfg::stage_bytes;
        cstage = cstage == STAGES - 1 ? 0 : cstage + 1;
        bf16x8 af[2][MT], bfr[2][NT];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int n = wc * (16 * NT) + 16 * j;
                bfr[s][j] = p8_frag<BKM>(st + (MB + (n >> 6)) * P8_PART, n & 63, s, lane);
            }
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int m = wr * (16 * MT) + 16 * i;
                af[s][i] = p8_frag<true>(st + (m >> 6) * P8_PART, m & 63, s, lane);
            }
        }
        if (!BKM) {        // transposing reads are inline asm: wait for them by hand (rule 18)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[s][j], af[s][i], acc[i][j], 0, 0, 0);
    }

    // ---- epilogue: accumulators -> f32 staging image in LDS (the ring is dead), then every thread owns 8 consecutive columns of one
    // row per pass: 16-byte loads and stores of whole row segments (a lane of an accumulator tile owns 4 columns of a row: written
    // directly, a wave instruction touches 32-byte pieces of 16 rows -- 1.5-2x slower on bf16 outputs, gemm.hip measured the same)
    const int lr = lane & 15, g4 = 4 * (lane >> 4);
    constexpr int LD = Cfg::BN + 4;                                   // f32 row pitch: conflict-free for the 4-column accumulator writes
    float* cs = reinterpret_cast<float*>(smem);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // the stream's trailing out-of-range pieces have written their zeros
    __syncthreads();                                                  // every wave has left the K loop
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
            store4(cs + (wr * (16 * MT) + 16 * i + lr) * LD + wc * (16 * NT) + 16 * j + g4, acc[i][j]);
    __syncthreads();
    constexpr int CG = Cfg::BN / 8, RP = 256 / CG;                    // column groups of 8; rows per pass
    const int c8 = (threadIdx.x % CG) * 8, r0 = threadIdx.x / CG;
    const bool col_ok = c8 < nvalid;                                  // N % 8 == 0: a group is in or out as a whole
    f32x4 b0 = {0, 0, 0, 0}, b1 = {0, 0, 0, 0};
    if (e.bias && col_ok) {
        b0 = load4(e.bias + n0 + c8);
        b1 = load4(e.bias + n0 + c8 + 4);
    }
    f32x4 s0 = {0, 0, 0, 0}, s1 = {0, 0, 0, 0};
#pragma unroll
    for (int pass = 0; pass < Cfg::BM / RP; ++pass) {
        const int row = pass * RP + r0;
        if (!col_ok || row >= mvalid) continue;
        f32x4 v0 = load4(cs + row * LD + c8), v1 = load4(cs + row * LD + c8 + 4);
        epi_row8(e, (unsigned)(m0 + row), n0 + c8, v0, v1, b0, b1);   // v0 / v1 come back as stored
        s0 += v0;
        s1 += v1;
    }
    if (e.colpart) {       // column sums of this tile: the RP row-threads of a column group -- in-wave by shuffles, the four waves via LDS
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (CG == 8) { s0[j] += __shfl_xor(s0[j], 8, 64); s1[j] += __shfl_xor(s1[j], 8, 64); }
            s0[j] += __shfl_xor(s0[j], 16, 64); s0[j] += __shfl_xor(s0[j], 32, 64);
            s1[j] += __shfl_xor(s1[j], 16, 64); s1[j] += __shfl_xor(s1[j], 32, 64);
        }
        __syncthreads();                                              // the staging image has been read
        float* cp = reinterpret_cast<float*>(smem);                    // [4 waves][BN]
        if (lane < CG) {
            store4(cp + wid * Cfg::BN + c8, s0);
            store4(cp + wid * Cfg::BN + c8 + 4, s1);
        }
        __syncthreads();
        for (int c = threadIdx.x; c < nvalid; c += 256)
            e.colpart[(int64_t)tm * e.N + n0 + c] = ((cp[c] + cp[Cfg::BN + c]) + cp[2 * Cfg::BN + c]) + cp[3 * Cfg::BN + c];
    }
}

template <bool BKM, int NB, int STAGES, int MB>
static void sm_launch_one(const bf16_t* a, int64_t lda, const bf16_t* b, int64_t ldb, int nk, int tiles_m, int tiles_n, const EpiDev& e,
                          hipStream_t s) {
    static bool attr_done = false;
    const int lds = STAGES * SmCfg<NB, MB>::stage_bytes;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)gemm_sm_kernel<BKM, NB, STAGES, MB>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_done = true;
    }
    gemm_sm_kernel<BKM, NB, STAGES, MB><<<tiles_m * tiles_n, 256, lds, s>>>(a, lda, b, ldb, nk, tiles_m, tiles_n, e);
}

// mb x nb = tile (64 mb) x (64 nb): 1 x 1, 1 x 2 or 2 x 2; stages = 3 | 4.  A k-major; b_kmajor selects the B layout.
void vaw_sm_launch(int mb, int nb, int stages, int b_kmajor, int64_t M, int64_t N, int64_t K, const bf16_t* a, int64_t lda,
                   const bf16_t* b, int64_t ldb, const EpiDev& e, hipStream_t s) {
    const int tiles_m = (int)((M + 64 * mb - 1) / (64 * mb)), tiles_n = (int)((N + 64 * nb - 1) / (64 * nb)), nk = (int)(K / 64);
#define SM_GO(BKv, NBv, STv, MBv) sm_launch_one<BKv, NBv, STv, MBv>(a, lda, b, ldb, nk, tiles_m, tiles_n, e, s)
#define SM_GO_B(NBv, STv, MBv) do { if (b_kmajor) SM_GO(true, NBv, STv, MBv); else SM_GO(false, NBv, STv, MBv); } while (0)
    if (mb == 2) { if (stages == 3) SM_GO_B(2, 3, 2); else SM_GO_B(2, 4, 2); }
    else if (nb == 1) { if (stages == 3) SM_GO_B(1, 3, 1); else SM_GO_B(1, 4, 1); }
    else { if (stages == 3) SM_GO_B(2, 3, 1); else SM_GO_B(2, 4, 1); }
}
