// Probe of ds_read_b64_tr_b8 (gfx950): which LDS bytes does each lane receive?  The LDS image holds its own byte offsets
// (low byte in one launch, high byte in another), lane l passes the address pitch * (l & 15) + 8 * (l >> 4) -- row l & 15 of a
// [16][pitch] byte image, 8-byte column block l >> 4 -- and the 8 bytes it gets back are printed as offsets, i.e. as
// (row, column) of the source image.  Measurement aid for an mn-major fp8 fragment path; not part of the library.
//   hipcc --offload-arch=gfx950 tools/probes/tr_b8_probe.hip -o /tmp/tr_b8_probe && /tmp/tr_b8_probe [pitch]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

__global__ void probe(uint64_t* out, int mode, int pitch) {
    __shared__ __attribute__((aligned(16))) unsigned char img[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) img[i] = mode == 0 ? (unsigned char)(i & 0xff) : (unsigned char)(i >> 8);
    __syncthreads();
    const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)img + pitch * (threadIdx.x & 15) + 8 * (threadIdx.x >> 4);
    uint64_t v;
    asm volatile("ds_read_b64_tr_b8 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out[threadIdx.x] = v;
}

int main(int argc, char** argv) {
    const int pitch = argc > 1 ? atoi(argv[1]) : 64;
    uint64_t *d, lo[64], hi[64];
    hipMalloc(&d, 64 * 8);
    probe<<<1, 64>>>(d, 0, pitch);
    hipMemcpy(lo, d, 64 * 8, hipMemcpyDeviceToHost);
    probe<<<1, 64>>>(d, 1, pitch);
    hipMemcpy(hi, d, 64 * 8, hipMemcpyDeviceToHost);
    printf("pitch %d: lane -> 8 x (row, col) of the [16][pitch] image\n", pitch);
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d (row %2d, blk %d):", l, l & 15, l >> 4);
        for (int j = 0; j < 8; ++j) {
            const int off = (int)((lo[l] >> (8 * j)) & 0xff) | ((int)((hi[l] >> (8 * j)) & 0xff) << 8);
            printf(" (%2d,%3d)", off / pitch, off % pitch);
        }
        printf("\n");
    }
    return 0;
}
