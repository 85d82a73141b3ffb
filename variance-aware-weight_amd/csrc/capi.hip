// Error plumbing and version of libvaw_hip.so (the kernels' extern "C" entry points live next to them).
#include <stdarg.h>
#include <stdio.h>

#include "common.h"

static thread_local char g_err[512] = "";

void vaw_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int vaw_version(void) { return 100; }   // 0.1.0
extern "C" const char* vaw_last_error_string(void) { return g_err; }

// Measurement aid (tools/contention_bench.py): n_wgs workgroups that each take a whole CU (160 KiB of LDS) and idle there for
// `microseconds` -- a stand-in for a collective kernel running on another stream beside the compute kernels.
__global__ void __launch_bounds__(64) cu_hog_kernel(uint64_t ticks) {
    extern __shared__ char hog_lds[];
    if (threadIdx.x == 0) hog_lds[0] = 1;
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();       // 100 MHz
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}
extern "C" int vaw_debug_cu_hog(int n_wgs, int microseconds, vaw_stream stream) {
    VAW_CHECK_ARG(n_wgs > 0 && n_wgs <= 256 && microseconds > 0 && microseconds <= 2000000, "cu_hog: arguments");
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)cu_hog_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
    cu_hog_kernel<<<n_wgs, 64, 160 * 1024, (hipStream_t)stream>>>((uint64_t)microseconds * 100);
    VAW_CHECK_LAUNCH("cu_hog");
    return VAW_OK;
}

// One-time uploads of descriptor tables (vaw_wgrad_grouped, vaw_reduce_rows_batched, vaw_fp8_quantize_delayed_batched): the callers
// build the table in a reused pageable array, and an asynchronous copy from pageable memory is only guaranteed to have been STAGED
// when the call returns for small sizes -- a later call that rewrites the array could corrupt a table that is uploaded exactly once.
// The table is therefore copied into a pinned buffer of its own that is never reused (a few KiB per group, groups are built once per
// workspace) and the asynchronous copy reads from there: no host synchronisation, legal inside a stream capture.
#include <mutex>
#include <string.h>
#include <vector>
hipError_t vaw_upload_table(void* dev, const void* host, size_t bytes, hipStream_t s) {
    static std::mutex mu;
    static std::vector<void*> keep;
    void* pinned = nullptr;
    hipError_t rc = hipHostMalloc(&pinned, bytes, hipHostMallocDefault);
    if (rc != hipSuccess) return rc;
    memcpy(pinned, host, bytes);
    {
        std::lock_guard<std::mutex> g(mu);
        keep.push_back(pinned);
    }
    return hipMemcpyAsync(dev, pinned, bytes, hipMemcpyHostToDevice, s);
}
