// Probe: how many bytes per clock a CU takes into LDS from L2-resident panels, by path -- measurement only, not part of the library.
//   The persistent GEMM's main loop is bound by operand arrival: 64 KiB per K tile by LDS-DMA at 22-27 B/clk/CU under
//   load, 40 without the MFMAs.  Question: is that the LDS-DMA path's own limit, i.e. do `buffer_load_dwordx4` to REGISTERS +
//   `ds_write_b128` add ingest beside it, or do both share one limit upstream (TA / L1)?
//   Workgroup = 8 waves; per "K tile" every wave moves PIECES (8) pieces of 1 KiB (lane x 16 B) from a panel every CU reads in
//   lockstep (L2 hits, like the GEMM's operand panels) into a 2-stage LDS ring; REG of the 8 pieces go through registers.
//   Optional companions per K tile: FRAG x `ds_read_b128` per wave (the GEMM reads 24 KiB per wave and K tile = 24) and MFMA x
//   `v_mfma_f32_16x16x32_bf16` per wave (the GEMM issues 64) on registers.
//     hipcc -O3 --offload-arch=gfx950 tools/probes/ingest_probe.hip -o /tmp/ingest_probe && /tmp/ingest_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)0x80000000u, 0x00020000);
}
constexpr int PIECES = 8, STAGE = 8 * PIECES * 1024;       // 64 KiB per K tile and workgroup

template <int REG, int FRAG, int MFMA>
__global__ void __launch_bounds__(512)
ingest_kernel(const char* __restrict__ panel, int panel_tiles, int n_tiles, float* __restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];                 // 2 stages
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const __amdgpu_buffer_rsrc_t rs = rsrc(panel);
    const unsigned voff = (unsigned)(wid * PIECES * 1024 + lane * 16);
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    bf16x8 fa = {1, 1, 1, 1, 1, 1, 1, 1}, fb = fa;
    u32x4 held[REG > 0 ? REG : 1];
    auto issue = [&](int kt) {
        char* dst = smem + (kt & 1) * STAGE + wid * PIECES * 1024;
        const unsigned soff = (unsigned)((kt % panel_tiles) * STAGE);
#pragma unroll
        for (int p = 0; p < PIECES; ++p) {
            if (p < PIECES - REG) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(dst + p * 1024), 16, voff + p * 1024, soff, 0, 0);
            else held[p - (PIECES - REG)] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + p * 1024, soff, 0);
        }
    };
    issue(0);
    for (int kt = 0; kt < n_tiles; ++kt) {
        // the register pieces of tile kt -> LDS, then everything of tile kt has landed
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (REG > 0) {
            char* dst = smem + (kt & 1) * STAGE + wid * PIECES * 1024;
#pragma unroll
            for (int r = 0; r < REG; ++r) *reinterpret_cast<u32x4*>(dst + (PIECES - REG + r) * 1024 + lane * 16) = held[r];
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (kt + 1 < n_tiles) issue(kt + 1);
        const char* src = smem + (kt & 1) * STAGE;
#pragma unroll
        for (int f = 0; f < FRAG; ++f) {
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(src + ((f * 8 + wid) % (8 * PIECES)) * 1024 + lane * 16);
            fa[f & 7] += v[0];
        }
#pragma unroll
        for (int m = 0; m < MFMA; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[m & 3], 0, 0, 0);
        asm volatile("s_barrier" ::: "memory");          // the stage may be refilled
    }
    if (acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + (float)fa[0] == 12345.f) sink[threadIdx.x] = acc[0][0];
}

template <int REG, int FRAG, int MFMA>
static void run(const char* panel, int panel_tiles, float* sink, const char* what) {
    const int n_tiles = 4096, grid = 256;
    CHECK(hipFuncSetAttribute((const void*)ingest_kernel<REG, FRAG, MFMA>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    ingest_kernel<REG, FRAG, MFMA><<<grid, 512, 2 * STAGE>>>(panel, panel_tiles, 64, sink);
    CHECK(hipEventRecord(e0));
    ingest_kernel<REG, FRAG, MFMA><<<grid, 512, 2 * STAGE>>>(panel, panel_tiles, n_tiles, sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double clk = 2.4e9 * ms * 1e-3;            // nominal clocks
    printf("%-44s REG=%d FRAG=%2d MFMA=%2d  %8.1f us  %6.0f clk per K tile  %5.1f B/clk/CU\n", what, REG, FRAG, MFMA, ms * 1e3,
           clk / n_tiles, (double)STAGE * n_tiles / clk);
}

int main() {
    const int panel_tiles = 32;                         // 2 MiB panel: L2-resident, every CU reads the same bytes in lockstep
    char* panel;
    float* sink;
    CHECK(hipMalloc(&panel, (size_t)panel_tiles * STAGE));
    CHECK(hipMemset(panel, 1, (size_t)panel_tiles * STAGE));
    CHECK(hipMalloc(&sink, 4096));
    run<0, 0, 0>(panel, panel_tiles, sink, "DMA only");
    run<2, 0, 0>(panel, panel_tiles, sink, "6 DMA + 2 through registers");
    run<4, 0, 0>(panel, panel_tiles, sink, "4 DMA + 4 through registers");
    run<8, 0, 0>(panel, panel_tiles, sink, "all through registers");
    run<0, 24, 0>(panel, panel_tiles, sink, "DMA + fragment reads");
    run<2, 24, 0>(panel, panel_tiles, sink, "6 + 2, fragment reads");
    run<4, 24, 0>(panel, panel_tiles, sink, "4 + 4, fragment reads");
    run<0, 24, 64>(panel, panel_tiles, sink, "DMA + fragment reads + MFMAs");
    run<2, 24, 64>(panel, panel_tiles, sink, "6 + 2, fragment reads + MFMAs");
    run<4, 24, 64>(panel, panel_tiles, sink, "4 + 4, fragment reads + MFMAs");
    run<8, 24, 64>(panel, panel_tiles, sink, "all through registers, reads + MFMAs");
    run<0, 0, 64>(panel, panel_tiles, sink, "DMA + MFMAs");
    return 0;
}
