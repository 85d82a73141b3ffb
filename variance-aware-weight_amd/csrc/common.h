// Shared device helpers for the gfx950 kernels (wave64, CDNA4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vaw_hip.h"

typedef __bf16 bf16_t;
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define VAW_WAVE 64

// ---- error plumbing (host) -------------------------------------------------
void vaw_set_error(const char* fmt, ...);
#define VAW_CHECK_ARG(cond, ...)                 \
    do {                                         \
        if (!(cond)) {                           \
            vaw_set_error(__VA_ARGS__);          \
            return VAW_ERR_INVALID;              \
        }                                        \
    } while (0)
#define VAW_CHECK_LAUNCH(name)                                                       \
    do {                                                                             \
        hipError_t e_ = hipGetLastError();                                           \
        if (e_ != hipSuccess) {                                                      \
            vaw_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));     \
            return VAW_ERR_LAUNCH;                                                   \
        }                                                                            \
    } while (0)

// one-time upload of a descriptor table from a pinned copy that stays alive (capi.hip)
hipError_t vaw_upload_table(void* dev, const void* host, size_t bytes, hipStream_t s);

// ---- element access by activation dtype ------------------------------------
__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }

// 4-wide vector load/store of activations (16 B for f32, 8 B for bf16)
__device__ __forceinline__ f32x4 load4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 load4(const bf16_t* p) {
    bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
    f32x4 r = {(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    return r;
}
__device__ __forceinline__ void store4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ void store4(bf16_t* p, f32x4 v) {
    bf16x4 r = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
    *reinterpret_cast<bf16x4*>(p) = r;
}

// ---- wave / block reductions -------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---- fp8 output of the row kernels (fp8 mode of dit.py with delayed scaling: the tensor's only reader is the quantiser) ----
// four values -> the fp8 bytes vaw_fp8_quantize_delayed makes of their bf16 roundings (scale inv = 1 / state[0], saturating)
__device__ __forceinline__ unsigned fp8_word_of_bf16(f32x4 v, float inv, int e5m2, float& amax) {
    f32x4 r = {(float)(bf16_t)v[0], (float)(bf16_t)v[1], (float)(bf16_t)v[2], (float)(bf16_t)v[3]};
    amax = fmaxf(fmaxf(amax, fmaxf(fabsf(r[0]), fabsf(r[1]))), fmaxf(fabsf(r[2]), fabsf(r[3])));
    const float fm = e5m2 ? 57344.f : 448.f;
    r = r * inv;
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = __builtin_amdgcn_fmed3f(r[j], -fm, fm);
    unsigned w = 0;
    if (e5m2) { w = __builtin_amdgcn_cvt_pk_bf8_f32(r[0], r[1], w, false); w = __builtin_amdgcn_cvt_pk_bf8_f32(r[2], r[3], w, true); }
    else { w = __builtin_amdgcn_cvt_pk_fp8_f32(r[0], r[1], w, false); w = __builtin_amdgcn_cvt_pk_fp8_f32(r[2], r[3], w, true); }
    return w;
}
// a wave's max |x| -> the running max of the scaling state (integer atomic max on the bits; look first: it only grows)
__device__ __forceinline__ void fp8_amax_commit(float m, float* acc, int lane) {
    m = wave_max(m);
    if (lane == 0 && m > 0.f && !(m != m)) {
        unsigned* a = reinterpret_cast<unsigned*>(acc);
        const unsigned mb = __float_as_uint(m);
        if (mb > __hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(a, mb);
    }
}
// Sum over a whole block; `scratch` needs blockDim.x/64 floats of LDS. Result valid in all threads.
__device__ __forceinline__ float block_sum(float v, float* scratch) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) scratch[wid] = v;
    __syncthreads();
    float r = 0.f;
    for (int i = 0; i < nw; ++i) r += scratch[i];
    return r;
}

// ---- activations ---------------------------------------------------------------
__device__ __forceinline__ float silu_f(float x) { return x / (1.f + __expf(-x)); }
__device__ __forceinline__ float silu_grad_f(float x) {
    float s = 1.f / (1.f + __expf(-x));
    return s * (1.f + x * (1.f - s));
}
// GELU, tanh approximation (nn.GELU(approximate="tanh"), reference models/dit.py:129), written through
// 0.5(1+tanh(u)) == sigmoid(2u): one v_exp_f32 + one reciprocal instead of a libm tanhf call per element.
__device__ __forceinline__ float gelu_tanh_f(float x) {
    const float k0 = 0.7978845608028654f, k1 = 0.044715f;
    const float u2 = 2.f * k0 * (x + k1 * x * x * x);
    return x / (1.f + __expf(-u2));
}
__device__ __forceinline__ float gelu_tanh_grad_f(float x) {
    const float k0 = 0.7978845608028654f, k1 = 0.044715f;
    const float x2 = x * x;
    const float u2 = 2.f * k0 * (x + k1 * x * x2);
    const float s = 1.f / (1.f + __expf(-u2));
    return s + x * s * (1.f - s) * (2.f * k0 * (1.f + 3.f * k1 * x2));
}

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }
