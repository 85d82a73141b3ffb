// Prototype of the "next kernel" DESIGN.md §6 argues for -- measurement only, not part of the library.
//   256 x 256 tile, FOUR waves (2 x 2), wave tile 128 x 128: 256 f32 accumulators per lane (the register file of a SIMD that
//   holds one wave: arch VGPRs + AGPRs), 1.5x fewer LDS fragment bytes per MFMA than the 8-wave kernel, ONE barrier per K tile.
//   C[M][N] (bf16) = A[M][K] . B[N][K]^T, both k-major bf16, M, N % 256 == 0, K % 64 == 0.  Non-persistent (one tile per
//   workgroup): at K = 4096..8192 the prologue / epilogue are noise, the main-loop rate is what is being measured.
//     hipcc -O3 --offload-arch=gfx950 tools/probes/gemm_p4_probe.hip -o /tmp/gemm_p4_probe && /tmp/gemm_p4_probe [M N K]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

#define PART 8192            // 64 rows x 128 B
#define STAGE (8 * PART)     // A parts 0-3, B parts 0-3
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)0x80000000u, 0x00020000);
}
#ifndef P4_AUX
#define P4_AUX 0             // cache-policy bits of the DMA loads (1 = sc0, 2 = nt): measurement switch
#endif
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rs, char* dst, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)dst, 16, voff, soff, 0, P4_AUX);
}
// 16 rows x 32 k fragment of a k-major part: row r16 + (l & 15), k-step s, chunk swizzled by (row >> 1) & 7
__device__ __forceinline__ bf16x8 frag(const char* part, int r16, int s, int lane) {
    const int row = r16 + (lane & 15);
    const int chunk = (4 * s + (lane >> 4)) ^ ((row >> 1) & 7);
    return *reinterpret_cast<const bf16x8*>(part + row * 128 + (chunk << 4));
}

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
gemm_p4_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, bf16_t* __restrict__ C, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];     // 2 stages x 64 KiB
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wid >> 1, wc = wid & 1;
    const int tiles_n = N / 256;
    // XCD-friendly order: the 32 tiles of an XCD in a round form a 4 x 8 block (see gemm_p8_kernel.h)
    int tile = blockIdx.x;
    {
        const int G = gridDim.x, x = tile & 7, q = G >> 3;
        tile = x * q + (tile >> 3);      // G % 8 == 0 here
    }
    const int tiles_m = M / 256;
    int tm, tn;
    {
        const int gsz = 4 * tiles_n, gi = tile / gsz, within = tile - gi * gsz;
        const int rows = tiles_m - 4 * gi < 4 ? tiles_m - 4 * gi : 4;
        tn = within / rows;
        tm = 4 * gi + within - tn * rows;
    }
    const int64_t m0 = (int64_t)tm * 256, n0 = (int64_t)tn * 256;
    const int nk = K / 64;
    const __amdgpu_buffer_rsrc_t rs_a = rsrc(A + m0 * K), rs_b = rsrc(B + n0 * K);
    // piece j of part p (8 pieces of 1 KiB): rows 8 j + (lane >> 3); this wave issues pieces 2 wid, 2 wid + 1 of every part
    unsigned off[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = 8 * (2 * wid + j) + (lane >> 3);
        const int chunk = (lane & 7) ^ ((r >> 1) & 7);
        off[j] = (unsigned)(r * K * 2 + chunk * 16);
    }
    auto issue = [&](int kt, int stage) {
        char* st = smem + stage * STAGE;
        const unsigned so = (unsigned)(kt * 128);
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                dma16(rs_a, st + p * PART + (2 * wid + j) * 1024, off[j] + (unsigned)(p * 64 * K * 2), so);
                dma16(rs_b, st + (4 + p) * PART + (2 * wid + j) * 1024, off[j] + (unsigned)(p * 64 * K * 2), so);
            }
    };
    f32x4 acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0, 0, 0, 0};

    // Software pipeline: the fragments of the NEXT k-step are read from LDS while the 64 MFMAs of the current one run
    // (two fragment buffers of 64 VGPRs); the DMA of K tile kt + 2 is issued at the middle of K tile kt, when the barrier
    // there has shown that every wave is done with stage kt % 2.
    auto load_frags = [&](int stage, int ks, bf16x8 (&af)[8], bf16x8 (&bfr)[8]) {
        const char* st = smem + stage * STAGE;
        const char* ap = st + (wr * 2) * PART;                 // this wave's 128 A rows = parts 2 wr, 2 wr + 1
        const char* bp = st + (4 + wc * 2) * PART;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            af[i] = frag(ap + (i >> 2) * PART, 16 * (i & 3), ks, lane);
            bfr[i] = frag(bp + (i >> 2) * PART, 16 * (i & 3), ks, lane);
        }
    };
    // one of this wave's 16 pieces of a K tile: c = 2 part + j (part 0-3 = A, 4-7 = B)
    auto issue_piece = [&](int kt, int stage, int c) {
        char* st = smem + stage * STAGE;
        const int part = c >> 1, j = c & 1;
        dma16(part < 4 ? rs_a : rs_b, st + part * PART + (2 * wid + j) * 1024, off[j] + (unsigned)((part & 3) * 64 * K * 2), (unsigned)(kt * 128));
    };
    // the same 64 MFMAs with the DMA of K tile kt_dma spread between them, two pieces per 8 MFMAs: issued in one burst the
    // 64 wave-instructions of a K tile queue up in front of the texture addresser (16 cycles each) and the waves stall on issue
    auto mma_dma = [&](const bf16x8 (&af)[8], const bf16x8 (&bfr)[8], int kt_dma, int stage, bool on) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(bfr[j]), "v"(af[i]));
#ifdef P4_FINE
                if (on && j == 3) issue_piece(kt_dma, stage, 2 * i);          // one piece per 4 MFMAs instead of two per 8
                if (on && j == 7) issue_piece(kt_dma, stage, 2 * i + 1);
#endif
            }
#ifndef P4_FINE
            if (on) { issue_piece(kt_dma, stage, 2 * i); issue_piece(kt_dma, stage, 2 * i + 1); }
#endif
        }
    };
    auto mma = [&](const bf16x8 (&af)[8], const bf16x8 (&bfr)[8]) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j)
                // inline asm pins the accumulators to AGPRs and the fragments to arch VGPRs (left to itself the allocator mixes
                // both files and moves fragments through v_accvgpr_write: 200 extra VALU ops per K tile)
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(bfr[j]), "v"(af[i]));   // transposed: lane = row, 4 cols
    };
#ifdef P4_REGSTAGE
    // Variant: global -> VGPR -> LDS (ds_write_b128) instead of LDS-DMA.  The 16 pieces of K tile kt + 2 are loaded into 64
    // staging VGPRs during the second half of K tile kt (spread between the MFMAs) and written to the stage that K tile kt has
    // just released during the first half of K tile kt + 1.
    bf16x8 stg[16];
    auto g_load = [&](int kt, int c) {
        const int part = c >> 1, j = c & 1;
        const bf16_t* base = part < 4 ? A + m0 * K : B + n0 * K;
        const char* src = (const char*)base + off[j] + (size_t)((part & 3) * 64) * K * 2 + (size_t)kt * 128;
        stg[c] = *reinterpret_cast<const bf16x8*>(src);
    };
    auto l_store = [&](int stage, int c) {
        const int part = c >> 1, j = c & 1;
        *reinterpret_cast<bf16x8*>(smem + stage * STAGE + part * PART + (2 * wid + j) * 1024 + lane * 16) = stg[c];
    };
#endif
    bf16x8 af0[8], bf0[8], af1[8], bf1[8];
#ifdef P4_REGSTAGE
    for (int c = 0; c < 16; ++c) { g_load(0, c); }
    for (int c = 0; c < 16; ++c) { l_store(0, c); }
    if (nk > 1) { for (int c = 0; c < 16; ++c) g_load(1, c); }
    __syncthreads();
    load_frags(0, 0, af0, bf0);
    for (int kt = 0; kt < nk; ++kt) {
        const int stage = kt & 1;
        load_frags(stage, 1, af1, bf1);
        // first half: MFMAs of k-step 0, with the staged K tile kt + 1 going to the other stage (released at the barrier below of
        // the previous iteration)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(bf0[j]), "v"(af0[i]));
            if (kt + 1 < nk) { l_store(stage ^ 1, 2 * i); l_store(stage ^ 1, 2 * i + 1); }
        }
        __syncthreads();                                       // stage kt + 1 complete and visible; (kt, 1) fragments are in registers
        if (kt + 1 < nk) load_frags(stage ^ 1, 0, af0, bf0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(bf1[j]), "v"(af1[i]));
            if (kt + 2 < nk) { g_load(kt + 2, 2 * i); g_load(kt + 2, 2 * i + 1); }
        }
    }
#else
    issue(0, 0);
    if (nk > 1) issue(1, 1);
    if (nk > 1) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    load_frags(0, 0, af0, bf0);
    for (int kt = 0; kt < nk; ++kt) {
        const int stage = kt & 1;
        load_frags(stage, 1, af1, bf1);
        mma(af0, bf0);
        // K tile kt + 1 (this wave's 16 pieces, issued a whole K tile ago) has landed; my reads of stage kt are complete
#ifdef P4_NO_VMWAIT          // ablations (wrong results, timing only): how much the DMA wait / the barrier / the DMA itself cost
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
#ifndef P4_NO_BARRIER
        __builtin_amdgcn_s_barrier();
#endif
        if (kt + 1 < nk) load_frags(stage ^ 1, 0, af0, bf0);
#ifndef P4_NO_DMA
        mma_dma(af1, bf1, kt + 2, stage, kt + 2 < nk);
#else
        mma(af1, bf1);
#endif
    }
#endif
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");     // (the asm MFMAs are invisible to the compiler's hazard recogniser)
    // epilogue: lane (l & 15) = row within the 16-row tile, 4 consecutive columns 4 (l >> 4) .. of each 16-column tile
    const int64_t mrow = m0 + wr * 128 + (lane & 15);
    const int64_t ncol = n0 + wc * 128 + 4 * (lane >> 4);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bf16x4 r = {(bf16_t)acc[i][j][0], (bf16_t)acc[i][j][1], (bf16_t)acc[i][j][2], (bf16_t)acc[i][j][3]};
            *reinterpret_cast<bf16x4*>(C + (mrow + 16 * i) * N + ncol + 16 * j) = r;
        }
}

int main(int argc, char** argv) {
    const int M = argc > 3 ? atoi(argv[1]) : 4096, N = argc > 3 ? atoi(argv[2]) : 4096, K = argc > 3 ? atoi(argv[3]) : 4096;
    if (M % 256 || N % 256 || K % 64 || ((M / 256) * (N / 256)) % 8) { printf("need M, N %% 256, K %% 64, tiles %% 8\n"); return 1; }
    std::vector<bf16_t> ha((size_t)M * K), hb((size_t)N * K);
    uint32_t s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((int)((s >> 16) % 5) - 2); };
    for (auto& v : ha) v = (bf16_t)rnd();
    for (auto& v : hb) v = (bf16_t)rnd();
    bf16_t *a, *b, *c;
    CHECK(hipMalloc(&a, ha.size() * 2)); CHECK(hipMalloc(&b, hb.size() * 2)); CHECK(hipMalloc(&c, (size_t)M * N * 2));
    CHECK(hipMemcpy(a, ha.data(), ha.size() * 2, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(b, hb.data(), hb.size() * 2, hipMemcpyHostToDevice));
    const int lds = 2 * STAGE, grid = (M / 256) * (N / 256);
    CHECK(hipFuncSetAttribute((const void*)gemm_p4_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    for (int i = 0; i < 3; ++i) gemm_p4_kernel<<<grid, 256, lds>>>(a, b, c, M, N, K);
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 10;
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) gemm_p4_kernel<<<grid, 256, lds>>>(a, b, c, M, N, K);
    hipEventRecord(e1);
    CHECK(hipDeviceSynchronize());
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = 1e3 * ms / iters;
    printf("%dx%dx%d: %.1f us, %.1f TFLOP/s\n", M, N, K, us, 2.0 * M * N * K / us / 1e6);
    // spot check: 64 entries against a host dot product (small integers: exact)
    std::vector<bf16_t> hc((size_t)M * N);
    CHECK(hipMemcpy(hc.data(), c, hc.size() * 2, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int t = 0; t < 64; ++t) {
        const int m = (t * 977 + 13) % M, n = (t * 613 + 7) % N;
        float ref = 0.f;
        for (int k = 0; k < K; ++k) ref += (float)ha[(size_t)m * K + k] * (float)hb[(size_t)n * K + k];
        if ((float)(bf16_t)ref != (float)hc[(size_t)m * N + n]) { if (bad < 4) printf("mismatch at (%d,%d): %f vs %f\n", m, n, (float)hc[(size_t)m * N + n], ref); ++bad; }
    }
    printf(bad ? "FAILED %d / 64\n" : "check ok\n", bad);
    return bad != 0;
}
