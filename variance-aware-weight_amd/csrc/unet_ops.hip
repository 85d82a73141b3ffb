// UNet-side kernels (reference models/unet.py + tools/nn.py), all on NHWC ("pixels x channels") activations so
// that every conv is a GEMM over M = B*H*W pixel rows and the attention qkv rows are token-major:
//   GroupNorm32 (+ FiLM scale/shift, + SiLU) forward / backward      tools/nn.py:17-19,93-100, unet.py:236-256
//   im2col / col2im for conv3x3 pad 1 (round 1: explicit patch matrix, GEMM does the arithmetic)
//   2x2 average pool, nearest x2 upsample (+ their transposes), channel concat / split, NCHW <-> NHWC
// All HBM-bound, 4 channels per thread (16 B f32 / 8 B bf16), reductions in a fixed order (no float atomics).
#include <stdlib.h>

#include "common.h"

static inline int sgrid(int64_t work, int block) {
    int64_t g = (work + block - 1) / block;
    return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}
#define BY_DTYPE(dt, CALL)                      \
    if (dt == VAW_F32) { using T = float; CALL; } \
    else { using T = bf16_t; CALL; }
#define GRID_STRIDE(i, n) for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < (n); i += (int64_t)gridDim.x * blockDim.x)

// ---------------------------------------------------------------------------------------------
// GroupNorm32 (+FiLM, +SiLU) forward and backward as streaming passes over [B, HW, C].  Every pass uses one
// thread mapping: a block covers 64 channels x one chunk of 512 pixels of one sample; a thread owns a channel QUAD
// (8-byte bf16 / 16-byte f32 accesses) and one of 16 row groups, so all per-channel parameters (mean, rstd, gamma,
// beta, FiLM scale/shift, group sums) are loaded once per thread and the row loop is pure streaming.
//   sums    per-(sample, channel) partial sums per chunk -> tiny group kernel folds chunks and channels in a fixed
//           order (double accumulation for the variance)
//   apply   y = act(GN(x)*gamma+beta [*(1+scale)+shift])    /    dx = rstd*(dn1*gamma - S1/N - xhat*S2/N) (+ dx_add)
// ---------------------------------------------------------------------------------------------
#define GN_ROWS 512
// SiLU and its derivative through v_rcp_f32 (1 ulp) instead of an IEEE division (~10 more instructions per element): the
// GroupNorm passes are close enough to VALU-bound at 4 waves per SIMD for that to show (GN_FAST_SILU=0: the exact forms)
#ifndef GN_FAST_SILU
#define GN_FAST_SILU 1
#endif
__device__ __forceinline__ float gn_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float gn_silu(float x) { return GN_FAST_SILU ? x * gn_sigmoid(x) : silu_f(x); }
__device__ __forceinline__ float gn_silu_grad(float x) {
    if (!GN_FAST_SILU) return silu_grad_f(x);
    const float s = gn_sigmoid(x);
    return s * (1.f + x * (1.f - s));
}

struct GnQuad {   // per-thread constants for its 4 channels
    f32x4 mu, rs, ga, be, sc, sh;
};
__device__ __forceinline__ GnQuad gn_quad(const float* mean, const float* rstd, const float* gamma, const float* beta,
                                          const float* scale, const float* shift, int64_t film_ld, int b, int c, int C, int G) {
    GnQuad q;
    const int cg = C / G;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int g = (c + j) / cg;
        q.mu[j] = mean[b * G + g];
        q.rs[j] = rstd[b * G + g];
    }
    q.ga = load4(gamma + c);
    q.be = load4(beta + c);
    q.sc = scale ? load4(scale + (int64_t)b * film_ld + c) : f32x4{0, 0, 0, 0};
    q.sh = scale ? load4(shift + (int64_t)b * film_ld + c) : f32x4{0, 0, 0, 0};
    return q;
}

template <typename T>
__global__ void __launch_bounds__(256)
gn_fwd_sums_kernel(const T* __restrict__ x, int HW, int C, int B, int nchunk, float* __restrict__ part /* [2][nchunk][B][C] */) {
    __shared__ __attribute__((aligned(16))) float red[2][16][64];
    const int cq = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int c = blockIdx.x * 64 + cq * 4, b = blockIdx.y, chunk = blockIdx.z;
    f32x4 s = {0, 0, 0, 0}, q = {0, 0, 0, 0};
    if (c < C) {
        const int r0 = chunk * GN_ROWS, r1 = r0 + GN_ROWS < HW ? r0 + GN_ROWS : HW;
        const T* p = x + (int64_t)b * HW * C + c;
        for (int r = r0 + rg; r < r1; r += 16) {
            const f32x4 v = load4(p + (int64_t)r * C);
            s += v;
            q += v * v;
        }
    }
    store4(&red[0][rg][cq * 4], s);
    store4(&red[1][rg][cq * 4], q);
    __syncthreads();
    const int cl = threadIdx.x & 63, k = threadIdx.x >> 6;
    if (k < 2 && blockIdx.x * 64 + cl < C) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) t += red[k][g][cl];
        part[(((int64_t)k * nchunk + chunk) * B + b) * C + blockIdx.x * 64 + cl] = t;
    }
}

__global__ void gn_group_stats_kernel(const float* __restrict__ part, int nchunk, int B, int C, int G, int HW, float eps,
                                      float* __restrict__ mean, float* __restrict__ rstd) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * G) return;
    const int b = i / G, g = i % G, cg = C / G;
    double s = 0.0, q = 0.0;
    for (int ch = 0; ch < nchunk; ++ch)
        for (int j = 0; j < cg; ++j) {
            s += (double)part[(((int64_t)0 * nchunk + ch) * B + b) * C + g * cg + j];
            q += (double)part[(((int64_t)1 * nchunk + ch) * B + b) * C + g * cg + j];
        }
    const double n = (double)cg * HW;
    const double m = s / n;
    double var = q / n - m * m;
    if (var < 0.0) var = 0.0;
    mean[i] = (float)m;
    rstd[i] = (float)(1.0 / sqrt(var + (double)eps));
}

template <typename T>
__global__ void __launch_bounds__(256)
gn_apply_kernel(const T* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
                const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ scale,
                const float* __restrict__ shift, int64_t film_ld, int silu, T* __restrict__ y, int HW, int C, int G) {
    const int cq = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int c = blockIdx.x * 64 + cq * 4, b = blockIdx.y, chunk = blockIdx.z;
    if (c >= C) return;
    const GnQuad k = gn_quad(mean, rstd, gamma, beta, scale, shift, film_ld, b, c, C, G);
    const f32x4 a1 = k.rs * k.ga, b1 = k.be - k.mu * k.rs * k.ga;       // n1 = x*a1 + b1
    const int r0 = chunk * GN_ROWS, r1 = r0 + GN_ROWS < HW ? r0 + GN_ROWS : HW;
    const int64_t base = (int64_t)b * HW * C + c;
    for (int r = r0 + rg; r < r1; r += 16) {
        f32x4 n = load4(x + base + (int64_t)r * C) * a1 + b1;
        if (scale) n = n * (1.f + k.sc) + k.sh;
        if (silu) {
#pragma unroll
            for (int j = 0; j < 4; ++j) n[j] = gn_silu(n[j]);
        }
        store4(y + base + (int64_t)r * C, n);
    }
}

// Backward sums per (sample, channel): A = dn1*xhat, Bs = dn1, DS = dn2*n1, DH = dn2
//   (n1 = xhat*gamma+beta, n2 = FiLM(n1), dn2 = dout*act'(n2), dn1 = dn2*(1+scale))
template <typename T>
__global__ void __launch_bounds__(256)
gn_bwd_sums_kernel(const T* __restrict__ dout, const T* __restrict__ x, const float* __restrict__ mean,
                   const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ beta,
                   const float* __restrict__ scale, const float* __restrict__ shift, int64_t film_ld, int silu, int HW, int C,
                   int G, int B, int nchunk, float* __restrict__ part /* [4][nchunk][B][C] */) {
    __shared__ __attribute__((aligned(16))) float red[4][16][64];
    const int cq = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int c = blockIdx.x * 64 + cq * 4, b = blockIdx.y, chunk = blockIdx.z;
    f32x4 a = {0, 0, 0, 0}, bs = {0, 0, 0, 0}, ds = {0, 0, 0, 0}, dh = {0, 0, 0, 0};
    if (c < C) {
        const GnQuad k = gn_quad(mean, rstd, gamma, beta, scale, shift, film_ld, b, c, C, G);
        const int r0 = chunk * GN_ROWS, r1 = r0 + GN_ROWS < HW ? r0 + GN_ROWS : HW;
        const int64_t base = (int64_t)b * HW * C + c;
        for (int r = r0 + rg; r < r1; r += 16) {
            const f32x4 xh = (load4(x + base + (int64_t)r * C) - k.mu) * k.rs;
            const f32x4 n1 = xh * k.ga + k.be;
            const f32x4 n2 = scale ? n1 * (1.f + k.sc) + k.sh : n1;
            f32x4 dn2 = load4(dout + base + (int64_t)r * C);
            if (silu) {
#pragma unroll
                for (int j = 0; j < 4; ++j) dn2[j] *= gn_silu_grad(n2[j]);
            }
            const f32x4 dn1 = scale ? dn2 * (1.f + k.sc) : dn2;
            a += dn1 * xh;
            bs += dn1;
            ds += dn2 * n1;
            dh += dn2;
        }
    }
    store4(&red[0][rg][cq * 4], a);
    store4(&red[1][rg][cq * 4], bs);
    store4(&red[2][rg][cq * 4], ds);
    store4(&red[3][rg][cq * 4], dh);
    __syncthreads();
    const int cl = threadIdx.x & 63, kk = threadIdx.x >> 6;
    if (blockIdx.x * 64 + cl < C) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) t += red[kk][g][cl];
        part[(((int64_t)kk * nchunk + chunk) * B + b) * C + blockIdx.x * 64 + cl] = t;
    }
}

// fold chunks -> per (b,c) sums; per (b,g): S1 = sum_c gamma_c*Bs, S2 = sum_c gamma_c*A; dgamma/dbeta over samples;
// FiLM gradients into the rows of the emb_layers output gradient.  One thread per (b, c) for the folds (fixed order).
__global__ void gn_bwd_fold_kernel(const float* __restrict__ part, int nchunk, int B, int C, float* __restrict__ sums /* [4][B][C] */) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)4 * B * C) return;
    const int64_t k = i / ((int64_t)B * C), bc = i % ((int64_t)B * C);
    float t = 0.f;
    for (int ch = 0; ch < nchunk; ++ch) t += part[((k * nchunk + ch) * B) * C + bc];
    sums[i] = t;
}
// Three independent jobs in one launch, told apart by block index (each fully parallel; fixed summation order):
//   blocks [0, nb1)        S1, S2 per (sample, group)
//   blocks [nb1, nb1+nb2)  dgamma, dbeta: 16 channels per block, 16 lanes stride the batch, shuffle tree over them
//   blocks [nb1+nb2, ...)  FiLM gradients copied out per (sample, channel)
// nchunk > 0: `sums` is the chunk partials [4][nchunk][B][C] and every read folds the chunks in place, in the order
// gn_bwd_fold_kernel does (the flat path: one launch fewer per GroupNorm backward); nchunk == 0: `sums` is already [4][B][C]
struct GnSums {
    const float* p;
    int nchunk;
    int64_t BC;
    __device__ __forceinline__ float operator()(int k, int64_t bc) const {
        if (nchunk == 0) return p[k * BC + bc];
        float t = 0.f;
        for (int ch = 0; ch < nchunk; ++ch) t += p[((int64_t)k * nchunk + ch) * BC + bc];
        return t;
    }
};
__global__ void __launch_bounds__(256)
gn_bwd_group_kernel(const float* __restrict__ sums, int nchunk, const float* __restrict__ gamma, int B, int C, int G,
                    float* __restrict__ S1, float* __restrict__ S2, float* __restrict__ dgamma,
                    float* __restrict__ dbeta, float gbeta, float* __restrict__ dscale,
                    float* __restrict__ dshift, int64_t dfilm_ld, int nb1, int nb2) {
    const int cg = C / G;
    const int64_t BC = (int64_t)B * C;
    const GnSums sum{sums, nchunk, BC};
    int blk = blockIdx.x;
    if (blk < nb1) {
        const int i = blk * 256 + threadIdx.x;
        if (i >= B * G) return;
        const int b = i / G, g = i % G;
        float s1 = 0.f, s2 = 0.f;
        for (int j = 0; j < cg; ++j) {
            const int c = g * cg + j;
            s1 += gamma[c] * sum(1, (int64_t)b * C + c);
            s2 += gamma[c] * sum(0, (int64_t)b * C + c);
        }
        S1[i] = s1;
        S2[i] = s2;
        return;
    }
    blk -= nb1;
    if (blk < nb2) {
        const int c = blk * 16 + (threadIdx.x & 15), bl = threadIdx.x >> 4;     // 16 channels x 16 batch lanes
        float dg = 0.f, db = 0.f;
        if (c < C)
            for (int b = bl; b < B; b += 16) {
                dg += sum(0, (int64_t)b * C + c);
                db += sum(1, (int64_t)b * C + c);
            }
        // lanes of one channel sit 16 apart: in-wave tree over lane bits 4,5, then the 4 waves through LDS
        dg += __shfl_xor(dg, 16, 64); dg += __shfl_xor(dg, 32, 64);
        db += __shfl_xor(db, 16, 64); db += __shfl_xor(db, 32, 64);
        __shared__ float red[2][4][16];
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        if (lane < 16) { red[0][w][lane] = dg; red[1][w][lane] = db; }
        __syncthreads();
        if (threadIdx.x < 16 && c < C) {
            const float tg = (red[0][0][threadIdx.x] + red[0][1][threadIdx.x]) + (red[0][2][threadIdx.x] + red[0][3][threadIdx.x]);
            const float tb = (red[1][0][threadIdx.x] + red[1][1][threadIdx.x]) + (red[1][2][threadIdx.x] + red[1][3][threadIdx.x]);
            dgamma[c] = (gbeta != 0.f ? gbeta * dgamma[c] : 0.f) + tg;
            dbeta[c] = (gbeta != 0.f ? gbeta * dbeta[c] : 0.f) + tb;
        }
        return;
    }
    blk -= nb2;
    const int64_t i = (int64_t)blk * 256 + threadIdx.x;
    if (dscale && i < BC) {
        const int64_t b = i / C, c = i % C;
        dscale[b * dfilm_ld + c] = sum(2, i);
        dshift[b * dfilm_ld + c] = sum(3, i);
    }
}

template <typename T>
__global__ void __launch_bounds__(256)
gn_bwd_apply_kernel(const T* __restrict__ dout, const T* __restrict__ x, const float* __restrict__ mean,
                    const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ beta,
                    const float* __restrict__ scale, const float* __restrict__ shift, int64_t film_ld, int silu,
                    const float* __restrict__ S1, const float* __restrict__ S2, const T* __restrict__ dx_add,
                    T* __restrict__ dx, int HW, int C, int G) {
    const int cq = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int c = blockIdx.x * 64 + cq * 4, b = blockIdx.y, chunk = blockIdx.z;
    if (c >= C) return;
    const GnQuad k = gn_quad(mean, rstd, gamma, beta, scale, shift, film_ld, b, c, C, G);
    const int cg = C / G;
    const float invn = 1.f / ((float)cg * HW);
    f32x4 t1, t2;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int g = (c + j) / cg;
        t1[j] = S1[b * G + g] * invn;
        t2[j] = S2[b * G + g] * invn;
    }
    const int r0 = chunk * GN_ROWS, r1 = r0 + GN_ROWS < HW ? r0 + GN_ROWS : HW;
    const int64_t base = (int64_t)b * HW * C + c;
    for (int r = r0 + rg; r < r1; r += 16) {
        const int64_t e = base + (int64_t)r * C;
        const f32x4 xh = (load4(x + e) - k.mu) * k.rs;
        const f32x4 n1 = xh * k.ga + k.be;
        const f32x4 n2 = scale ? n1 * (1.f + k.sc) + k.sh : n1;
        f32x4 dn2 = load4(dout + e);
        if (silu) {
#pragma unroll
            for (int j = 0; j < 4; ++j) dn2[j] *= gn_silu_grad(n2[j]);
        }
        const f32x4 dn1 = scale ? dn2 * (1.f + k.sc) : dn2;
        f32x4 r4 = k.rs * (dn1 * k.ga - t1 - xh * t2);
        if (dx_add) r4 += load4(dx_add + e);
        store4(dx + e, r4);
    }
}

// ---------------------------------------------------------------------------------------------
// The same four passes for bf16 with C % 8 == 0 on a FLAT mapping (the recipe that took the AdamW and row kernels from
// 4-5 to 5-6.5 TB/s): a thread owns one channel OCTET (16-byte accesses) and every rpi-th row, nt = a multiple of C/8 lanes
// are live, so one step of a workgroup is nt x 16 CONTIGUOUS bytes; U steps are in flight per lane and stream before the
// first is consumed; non-temporal accesses for everything that is not read again soon (the forward sums pass leaves x in
// the caches for the apply pass, which walks the chunks in the opposite order so that it starts on the freshest ones).
// Chunks (GN_ROWS rows), workspace layouts and the small fold / group kernels are those of the quad-mapped kernels above,
// which stay for f32, C % 8 != 0 and VAW_GN_FLAT=0.  Backward sums: with d = dout act'(n2) only sd = sum d and
// sx = sum d (x - mu) are accumulated; A, Bs, DS, DH are formed from them when the chunk's sums are written.
// ---------------------------------------------------------------------------------------------
#define GNS_NT 256
typedef unsigned gns_u32x4 __attribute__((ext_vector_type(4)));
struct GnsGeom {
    int C8, nt, rpi, rows;     // rows per chunk
};
// chunk rows: GN_ROWS, or less while that leaves fewer than two workgroups per CU (0: use the quad-mapped kernels)
static int g_gn_flat = -1;       // vaw_debug_gn_flat: -1 by shape (VAW_GN_FLAT=0 switches the flat kernels off), 0 never, 1 always (tests)
extern "C" void vaw_debug_gn_flat(int mode) { g_gn_flat = mode; }
static inline int gns_rows(int B, int HW) {
    if (g_gn_flat == 0) return 0;
    for (int rows = GN_ROWS; rows >= GN_ROWS / 4; rows /= 2)
        if ((int64_t)B * ((HW + rows - 1) / rows) >= 512) return rows;
    return g_gn_flat == 1 ? GN_ROWS / 4 : 0;
}
static inline GnsGeom gns_geom(int C, int rows) {
    GnsGeom g;
    g.C8 = C / 8;
    g.nt = (GNS_NT / g.C8) * g.C8;
    g.rpi = g.nt / g.C8;
    g.rows = rows;
    return g;
}
__device__ __forceinline__ gns_u32x4 gns_ld(const bf16_t* p) { return *reinterpret_cast<const gns_u32x4*>(p); }
__device__ __forceinline__ gns_u32x4 gns_ld_nt(const bf16_t* p) { return __builtin_nontemporal_load(reinterpret_cast<const gns_u32x4*>(p)); }
__device__ __forceinline__ void gns_unpack(gns_u32x4 v, float (&f)[8]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        f[2 * k] = __uint_as_float(v[k] << 16);
        f[2 * k + 1] = __uint_as_float(v[k] & 0xffff0000u);
    }
}
__device__ __forceinline__ void gns_st_nt(bf16_t* p, const float (&f)[8]) {
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (bf16_t)f[j];
    __builtin_nontemporal_store(v, reinterpret_cast<bf16x8*>(p));
}
// per-thread sums of 8 channels x 2 quantities -> per-channel sums of the chunk in chs[2][C] (C <= 2048)
__device__ __forceinline__ void gns_fold(const float (&s)[8], const float (&q)[8], float* red, float* chs, int C, int rpi, int rl, int c0,
                                         bool live) {
    if (live) {
        float* w0 = red + rl * C + c0;
        float* w1 = red + (rpi + rl) * C + c0;
        *reinterpret_cast<f32x4*>(w0) = f32x4{s[0], s[1], s[2], s[3]};
        *reinterpret_cast<f32x4*>(w0 + 4) = f32x4{s[4], s[5], s[6], s[7]};
        *reinterpret_cast<f32x4*>(w1) = f32x4{q[0], q[1], q[2], q[3]};
        *reinterpret_cast<f32x4*>(w1 + 4) = f32x4{q[4], q[5], q[6], q[7]};
    }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * C; c += GNS_NT) {
        const int k = c >= C ? 1 : 0, cc = c - k * C;
        const float* p = red + k * rpi * C + cc;
        float t = 0.f;
        for (int r = 0; r < rpi; ++r) t += p[r * C];
        chs[c] = t;
    }
    __syncthreads();
}

template <int U>
__global__ void __launch_bounds__(GNS_NT)
gns_fwd_sums_kernel(const bf16_t* __restrict__ x, int HW, int C, int B, int nchunk, GnsGeom gm, float* __restrict__ part /* [2][nchunk][B][C] */) {
    __shared__ __attribute__((aligned(16))) float red[2 * GNS_NT * 8];
    __shared__ float chs[2 * 2048];
    const int t = threadIdx.x, b = blockIdx.x / nchunk, chunk = blockIdx.x - b * nchunk;
    const bool live = t < gm.nt;
    const int oct = live ? t % gm.C8 : 0, rl = live ? t / gm.C8 : 0, c0 = oct * 8;
    const int r0 = chunk * gm.rows, r1 = r0 + gm.rows < HW ? r0 + gm.rows : HW;
    const bf16_t* xs = x + (int64_t)b * HW * C;
    float s[8], q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = q[j] = 0.f;
    for (int rb = r0 + rl; rb < r1; rb += U * gm.rpi) {
        gns_u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int r = rb + u * gm.rpi;
            r = r < r1 ? r : r1 - 1;
            v[u] = gns_ld(xs + (unsigned)(r * C + c0));
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool ok = live && rb + u * gm.rpi < r1;
            float f[8];
            gns_unpack(v[u], f);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float a = ok ? f[j] : 0.f;
                s[j] += a;
                q[j] += a * a;
            }
        }
    }
    gns_fold(s, q, red, chs, C, gm.rpi, rl, c0, live);
    for (int c = t; c < 2 * C; c += GNS_NT) {
        const int k = c >= C ? 1 : 0, cc = c - k * C;
        part[(((int64_t)k * nchunk + chunk) * B + b) * C + cc] = chs[c];
    }
}

template <int U, bool SILU, bool FILM>
__global__ void __launch_bounds__(GNS_NT)
gns_apply_kernel(const bf16_t* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
                 const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ scale,
                 const float* __restrict__ shift, int64_t film_ld, bf16_t* __restrict__ y, int HW, int C, int G, int nchunk, GnsGeom gm) {
    const int item = gridDim.x - 1 - blockIdx.x;             // the sums pass went 0 .. n-1: start on what it read last
    const int t = threadIdx.x, b = item / nchunk, chunk = item - b * nchunk;
    if (t >= gm.nt) return;
    const int oct = t % gm.C8, rl = t / gm.C8, c0 = oct * 8, cg = C / G;
    const int r0 = chunk * gm.rows, r1 = r0 + gm.rows < HW ? r0 + gm.rows : HW;
    const bf16_t* xs = x + (int64_t)b * HW * C;
    bf16_t* ys = y + (int64_t)b * HW * C;
    float a1[8], b1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int g = (c0 + j) / cg;
        const float mu = mean[b * G + g], rs = rstd[b * G + g], ga = gamma[c0 + j];
        a1[j] = rs * ga;
        b1[j] = beta[c0 + j] - mu * rs * ga;
        if (FILM) {                                          // (x a + b)(1 + scale) + shift as one multiply-add
            const float sc = 1.f + scale[(int64_t)b * film_ld + c0 + j];
            a1[j] *= sc;
            b1[j] = b1[j] * sc + shift[(int64_t)b * film_ld + c0 + j];
        }
    }
    for (int rb = r0 + rl; rb < r1; rb += U * gm.rpi) {
        gns_u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int r = rb + u * gm.rpi;
            r = r < r1 ? r : r1 - 1;
            v[u] = gns_ld_nt(xs + (unsigned)(r * C + c0));
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int r = rb + u * gm.rpi;
            float f[8], o[8];
            gns_unpack(v[u], f);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float n = f[j] * a1[j] + b1[j];
                if (SILU) n = gn_silu(n);
                o[j] = n;
            }
            if (r < r1) gns_st_nt(ys + (unsigned)(r * C + c0), o);
        }
    }
}

// n2 = x P + Q,  P = rstd gamma (1+scale),  Q = (beta - mu rstd gamma)(1+scale) + shift
template <int U, bool SILU, bool FILM>
__global__ void __launch_bounds__(GNS_NT)
gns_bwd_sums_kernel(const bf16_t* __restrict__ dout, const bf16_t* __restrict__ x, const float* __restrict__ mean,
                    const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ beta,
                    const float* __restrict__ scale, const float* __restrict__ shift, int64_t film_ld, int HW, int C, int G, int B,
                    int nchunk, GnsGeom gm, float* __restrict__ part /* [4][nchunk][B][C] */) {
    __shared__ __attribute__((aligned(16))) float red[2 * GNS_NT * 8];
    __shared__ float chs[2 * 2048];
    const int t = threadIdx.x, b = blockIdx.x / nchunk, chunk = blockIdx.x - b * nchunk;
    const bool live = t < gm.nt;
    const int oct = live ? t % gm.C8 : 0, rl = live ? t / gm.C8 : 0, c0 = oct * 8, cg = C / G;
    const int r0 = chunk * gm.rows, r1 = r0 + gm.rows < HW ? r0 + gm.rows : HW;
    const bf16_t *xs = x + (int64_t)b * HW * C, *ds = dout + (int64_t)b * HW * C;
    float P[8], Q[8], mu[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int g = (c0 + j) / cg;
        mu[j] = mean[b * G + g];
        const float rs = rstd[b * G + g], ga = gamma[c0 + j];
        const float s1 = FILM ? 1.f + scale[(int64_t)b * film_ld + c0 + j] : 1.f;
        P[j] = rs * ga * s1;
        Q[j] = (beta[c0 + j] - mu[j] * rs * ga) * s1 + (FILM ? shift[(int64_t)b * film_ld + c0 + j] : 0.f);
    }
    float sd[8], sx[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) sd[j] = sx[j] = 0.f;
    for (int rb = r0 + rl; rb < r1; rb += U * gm.rpi) {
        gns_u32x4 xv[U], dv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int r = rb + u * gm.rpi;
            r = r < r1 ? r : r1 - 1;
            xv[u] = gns_ld(xs + (unsigned)(r * C + c0));
            dv[u] = gns_ld(ds + (unsigned)(r * C + c0));
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool ok = live && rb + u * gm.rpi < r1;
            float xf[8], df[8];
            gns_unpack(xv[u], xf);
            gns_unpack(dv[u], df);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float d = df[j];
                if (SILU) d *= gn_silu_grad(xf[j] * P[j] + Q[j]);
                d = ok ? d : 0.f;
                sd[j] += d;
                sx[j] += d * (xf[j] - mu[j]);
            }
        }
    }
    gns_fold(sd, sx, red, chs, C, gm.rpi, rl, c0, live);
    // A = (1+scale) rstd sx,  Bs = (1+scale) sd,  DS = gamma rstd sx + beta sd,  DH = sd
    const int64_t plane = (int64_t)nchunk * B * C;
    for (int c = t; c < C; c += GNS_NT) {
        const float d0 = chs[c], x0 = chs[C + c];
        const float rs = rstd[b * G + c / cg];
        const float s1 = FILM ? 1.f + scale[(int64_t)b * film_ld + c] : 1.f;
        float* o = part + ((int64_t)chunk * B + b) * C + c;
        o[0] = s1 * (rs * x0);
        o[plane] = s1 * d0;
        o[2 * plane] = gamma[c] * (rs * x0) + beta[c] * d0;
        o[3 * plane] = d0;
    }
}

// dx = rstd (dn1 gamma - S1/N - xhat S2/N) + dx_add  =  P d - K3 x - K2 (+ dx_add),  K3 = rstd^2 S2/N,  K2 = rstd S1/N - K3 mu
template <int U, bool SILU, bool FILM, bool ADD>
__global__ void __launch_bounds__(GNS_NT)
gns_bwd_apply_kernel(const bf16_t* __restrict__ dout, const bf16_t* __restrict__ x, const float* __restrict__ mean,
                     const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ beta,
                     const float* __restrict__ scale, const float* __restrict__ shift, int64_t film_ld,
                     const float* __restrict__ S1, const float* __restrict__ S2, const bf16_t* __restrict__ dx_add,
                     bf16_t* __restrict__ dx, int HW, int C, int G, int nchunk, GnsGeom gm) {
    const int item = gridDim.x - 1 - blockIdx.x;             // the sums pass went 0 .. n-1: start on what it read last
    const int t = threadIdx.x, b = item / nchunk, chunk = item - b * nchunk;
    if (t >= gm.nt) return;
    const int oct = t % gm.C8, rl = t / gm.C8, c0 = oct * 8, cg = C / G;
    const int r0 = chunk * gm.rows, r1 = r0 + gm.rows < HW ? r0 + gm.rows : HW;
    const int64_t sample = (int64_t)b * HW * C;
    const bf16_t *xs = x + sample, *ds = dout + sample, *as = dx_add + sample;
    bf16_t* os = dx + sample;
    const float invn = 1.f / ((float)cg * HW);
    float P[8], Q[8], K2[8], K3[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int g = (c0 + j) / cg;
        const float mu = mean[b * G + g], rs = rstd[b * G + g], ga = gamma[c0 + j];
        const float s1 = FILM ? 1.f + scale[(int64_t)b * film_ld + c0 + j] : 1.f;
        P[j] = rs * ga * s1;
        Q[j] = (beta[c0 + j] - mu * rs * ga) * s1 + (FILM ? shift[(int64_t)b * film_ld + c0 + j] : 0.f);
        K3[j] = rs * rs * (S2[b * G + g] * invn);
        K2[j] = rs * (S1[b * G + g] * invn) - K3[j] * mu;
    }
    for (int rb = r0 + rl; rb < r1; rb += U * gm.rpi) {
        gns_u32x4 xv[U], dv[U], av[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int r = rb + u * gm.rpi;
            r = r < r1 ? r : r1 - 1;
            xv[u] = gns_ld_nt(xs + (unsigned)(r * C + c0));
            dv[u] = gns_ld_nt(ds + (unsigned)(r * C + c0));
            if (ADD) av[u] = gns_ld_nt(as + (unsigned)(r * C + c0));
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int r = rb + u * gm.rpi;
            float xf[8], df[8], af[8], o[8];
            gns_unpack(xv[u], xf);
            gns_unpack(dv[u], df);
            if (ADD) gns_unpack(av[u], af);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float d = df[j];
                if (SILU) d *= gn_silu_grad(xf[j] * P[j] + Q[j]);
                float v = P[j] * d - K3[j] * xf[j] - K2[j];
                if (ADD) v += af[j];
                o[j] = v;
            }
            if (r < r1) gns_st_nt(os + (unsigned)(r * C + c0), o);
        }
    }
}

#include "groupnorm_coop.h"

// ---------------------------------------------------------------------------------------------
// conv3x3 (stride 1, pad 1) patch matrix: col[m, tap*C + c] = x[pixel(m) + (tap/3-1, tap%3-1), c]  (0 outside)
// col2im is its transpose as a GATHER: dx[p, c] = sum_tap dcol[p - shift(tap), tap*C + c]
// ---------------------------------------------------------------------------------------------
template <typename T, int V>
__global__ void im2col3x3_kernel(const T* __restrict__ x, T* __restrict__ col, int B, int H, int W, int C) {
    const int64_t total = (int64_t)B * H * W * 9 * (C / V);
    GRID_STRIDE(i, total) {
        const int cv = (int)(i % (C / V));
        int64_t r = i / (C / V);
        const int tap = (int)(r % 9);
        const int64_t m = r / 9;
        const int w = (int)(m % W), h = (int)((m / W) % H);
        const int hh = h + tap / 3 - 1, ww = w + tap % 3 - 1;
        T* dst = col + (m * 9 + tap) * C + cv * V;
        const bool in = hh >= 0 && hh < H && ww >= 0 && ww < W;
        const T* src = x + (m + (int64_t)(tap / 3 - 1) * W + (tap % 3 - 1)) * C + cv * V;
        if (V == 4) {
            store4(dst, in ? load4(src) : f32x4{0, 0, 0, 0});
        } else {
            dst[0] = in ? src[0] : from_f32<T>(0.f);
        }
    }
}

template <typename T, int V>
__global__ void col2im3x3_kernel(const T* __restrict__ dcol, T* __restrict__ dx, int B, int H, int W, int C) {
    const int64_t total = (int64_t)B * H * W * (C / V);
    GRID_STRIDE(i, total) {
        const int cv = (int)(i % (C / V));
        const int64_t p = i / (C / V);
        const int w = (int)(p % W), h = (int)((p / W) % H);
        f32x4 acc = {0, 0, 0, 0};
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dh = tap / 3 - 1, dw = tap % 3 - 1;
            const int hh = h - dh, ww = w - dw;     // the pixel whose patch holds p at position `tap`
            if (hh >= 0 && hh < H && ww >= 0 && ww < W) {
                const T* src = dcol + ((p - (int64_t)dh * W - dw) * 9 + tap) * C + cv * V;
                if (V == 4) acc += load4(src);
                else acc[0] += to_f32(src[0]);
            }
        }
        if (V == 4) store4(dx + p * C + cv * 4, acc);
        else dx[p * C + cv] = from_f32<T>(acc[0]);
    }
}

// ---------------------------------------------------------------------------------------------
// conv3x3 weight gradient when one side has <= 4 channels (the 3-channel stem and output convs): a GEMM would
// have a 3- or 27-wide dimension.  One thread per channel of the WIDE side keeps its 9*S accumulators (S = small
// channel count) in registers and walks a chunk of pixels; chunk partials are folded in a fixed order.
//   SMALL_IN : dW[co][tap][ci] = sum_m dy[m][co] * x[nbr(m,tap)][ci],  thread = co, ci < S
//   !SMALL_IN: same sum, thread = ci, co < S
// ---------------------------------------------------------------------------------------------
#define WG_CHUNK 256
template <typename T, int S, bool SMALL_IN>
__global__ void __launch_bounds__(256)
conv3x3_wgrad_small_kernel(const T* __restrict__ dy, const T* __restrict__ x, float* __restrict__ part, int B, int H, int W,
                           int Cw /* wide channel count */) {
    // Per workgroup: WG_CHUNK pixels x 256 wide channels.  What every thread needs of a pixel -- the 3x3 x S patch of the
    // narrow input (SMALL_IN) or the S narrow output gradients plus the tap validity (SMALL_OUT) -- is gathered ONCE into
    // LDS by thread = pixel and then read back as broadcasts, instead of 256 threads re-loading the same scalars.
    constexpr int PS = SMALL_IN ? (9 * S + 3) / 4 * 4 : 4;
    __shared__ __attribute__((aligned(16))) float pix[WG_CHUNK][PS];
    __shared__ unsigned short tapmask[WG_CHUNK];             // bit t: tap t of this pixel lies inside the image
    const int c = blockIdx.y * 256 + threadIdx.x;
    const int64_t M = (int64_t)B * H * W;
    const int64_t m0 = (int64_t)blockIdx.x * WG_CHUNK, m1 = m0 + WG_CHUNK < M ? m0 + WG_CHUNK : M;
    {
        const int64_t m = m0 + threadIdx.x;
        const int w = (int)(m % W), h = (int)((m / W) % H);
        unsigned mask = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int hh = h + t / 3 - 1, ww = w + t % 3 - 1;
            const bool ok = m < M && hh >= 0 && hh < H && ww >= 0 && ww < W;
            mask |= ok ? (1u << t) : 0u;
            if (SMALL_IN) {
                const T* xp = x + (m + (int64_t)(t / 3 - 1) * W + (t % 3 - 1)) * S;
#pragma unroll
                for (int j = 0; j < S; ++j) pix[threadIdx.x][t * S + j] = ok ? to_f32(xp[j]) : 0.f;
            }
        }
        if (SMALL_IN) {
#pragma unroll
            for (int k = 9 * S; k < PS; ++k) pix[threadIdx.x][k] = 0.f;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) pix[threadIdx.x][j] = (j < S && m < M) ? to_f32(dy[m * S + j]) : 0.f;
        }
        tapmask[threadIdx.x] = (unsigned short)mask;
    }
    __syncthreads();
    float acc[9][S];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < S; ++j) acc[t][j] = 0.f;
    if (c < Cw) {
        for (int64_t m = m0; m < m1; ++m) {
            const int lm = (int)(m - m0);
            if (SMALL_IN) {
                const float g = to_f32(dy[m * Cw + c]);
#pragma unroll
                for (int k4 = 0; k4 < PS / 4; ++k4) {
                    const f32x4 pv = load4(&pix[lm][4 * k4]);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int k = 4 * k4 + u;
                        if (k < 9 * S) acc[k / S][k % S] += g * pv[u];       // padded taps hold zeros
                    }
                }
            } else {
                const f32x4 g = load4(&pix[lm][0]);
                const unsigned mask = tapmask[lm];
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    if (mask & (1u << t)) {                  // uniform across the workgroup: same pixel for every thread
                        const float xv = to_f32(x[(m + (int64_t)(t / 3 - 1) * W + (t % 3 - 1)) * Cw + c]);
#pragma unroll
                        for (int j = 0; j < S; ++j) acc[t][j] += g[j] * xv;
                    }
                }
            }
        }
        // partial layout = the weight layout [Co][9][Ci], one slab per chunk
        float* out = part + (int64_t)blockIdx.x * 9 * S * Cw;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int j = 0; j < S; ++j) {
                if (SMALL_IN) out[((int64_t)c * 9 + t) * S + j] = acc[t][j];
                else out[((int64_t)j * 9 + t) * Cw + c] = acc[t][j];
            }
    }
}
__global__ void fold_slabs_kernel(const float* __restrict__ part, int64_t nslab, int64_t n, float* __restrict__ out, float beta) {
    __shared__ float fold[8][33];
    const int cl = threadIdx.x & 31, grp = threadIdx.x >> 5;   // 32 outputs x 8 slab groups
    const int64_t i = (int64_t)blockIdx.x * 32 + cl;
    float acc = 0.f;
    if (i < n)
        for (int64_t sidx = grp; sidx < nslab; sidx += 8) acc += part[sidx * n + i];
    fold[grp][cl] = acc;
    __syncthreads();
    if (grp == 0 && i < n) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) t += fold[g][cl];
        out[i] = (beta != 0.f ? beta * out[i] : 0.f) + t;
    }
}

extern "C" int64_t vaw_conv3x3_wgrad_small_workspace_floats(int B, int H, int W, int Ci, int Co) {
    const int64_t M = (int64_t)B * H * W;
    return ((M + WG_CHUNK - 1) / WG_CHUNK) * 9 * Ci * Co;
}

extern "C" int vaw_conv3x3_wgrad_small(vaw_dtype dt, const void* dy, const void* x, float* dw, float beta, int B, int H, int W,
                                       int Ci, int Co, float* workspace, int64_t workspace_floats, vaw_stream stream) {
    VAW_CHECK_ARG(B > 0 && H > 0 && W > 0 && Ci > 0 && Co > 0 && (Ci <= 4 || Co <= 4), "conv3x3_wgrad_small: needs Ci<=4 or Co<=4");
    VAW_CHECK_ARG(workspace && workspace_floats >= vaw_conv3x3_wgrad_small_workspace_floats(B, H, W, Ci, Co), "conv3x3_wgrad_small: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int64_t M = (int64_t)B * H * W;
    const int nchunk = (int)((M + WG_CHUNK - 1) / WG_CHUNK);
    const bool small_in = Ci <= 4;
    const int S = small_in ? Ci : Co, Cw = small_in ? Co : Ci;
    dim3 grid(nchunk, ceil_div(Cw, 256));
#define WG_LAUNCH(Sv)                                                                                                          \
    if (small_in) { BY_DTYPE(dt, (conv3x3_wgrad_small_kernel<T, Sv, true><<<grid, 256, 0, s>>>((const T*)dy, (const T*)x, workspace, B, H, W, Cw))); } \
    else { BY_DTYPE(dt, (conv3x3_wgrad_small_kernel<T, Sv, false><<<grid, 256, 0, s>>>((const T*)dy, (const T*)x, workspace, B, H, W, Cw))); }
    switch (S) {
        case 1: WG_LAUNCH(1) break;
        case 2: WG_LAUNCH(2) break;
        case 3: WG_LAUNCH(3) break;
        default: WG_LAUNCH(4) break;
    }
    const int64_t n = 9LL * Ci * Co;
    fold_slabs_kernel<<<ceil_div(n, 32), 256, 0, s>>>(workspace, nchunk, n, dw, beta);
    VAW_CHECK_LAUNCH("conv3x3_wgrad_small");
    return VAW_OK;
}

// ---------------------------------------------------------------------------------------------
// conv3x3 with a narrow side (<= 4 channels): the 3-channel stem and the 3-channel output conv.  27 (or 36) taps
// per output are too short a K for the MFMA tile and the patch matrix would be pure HBM traffic, so these are direct
// f32-accumulating kernels: weights live in LDS as f32, activations are read once, outputs written in 16-byte pieces.
//   WIDE_OUT  in [M][S] -> out [M][Cw]; flip=0: forward (w = [Cw][9][S]);  flip=1: input gradient of a conv whose
//             OUTPUT is narrow: in = dy [M][S], out = dx [M][Cw], w = [S][9][Cw] read with the tap reversed
//   NARROW_OUT in [M][Cw] -> out [M][S], forward, w = [S][9][Cw]
// ---------------------------------------------------------------------------------------------
#define NW_PIX 2          // pixels per thread (the LDS weight reads are shared between them)
#define NW_BLOCK_PIX 256  // pixels per workgroup
template <typename T, int S>
__global__ void __launch_bounds__(256)
conv3x3_narrow_in_kernel(const T* __restrict__ in, const T* __restrict__ w, const float* __restrict__ bias, T* __restrict__ out,
                         int B, int H, int W, int Cw, int flip) {
    constexpr int PS = (9 * S + 3) / 4 * 4;          // patch row padded to whole 16-byte quads
    __shared__ float wl[9 * S][256 + 8];             // [tap*S + j][wide channel of this 256-chunk]
    __shared__ __attribute__((aligned(16))) float pat[NW_BLOCK_PIX][PS];   // the 3x3 x S patch of every pixel of the workgroup
    const int c0 = blockIdx.y * 256, cn = Cw - c0 < 256 ? Cw - c0 : 256;
    for (int i = threadIdx.x; i < 9 * S * cn; i += 256) {
        const int c = i % cn, kj = i / cn, tap = kj / S, j = kj % S;
        wl[kj][c] = flip ? to_f32(w[((int64_t)j * 9 + (8 - tap)) * Cw + c0 + c]) : to_f32(w[((int64_t)(c0 + c) * 9 + tap) * S + j]);
    }
    const int64_t M = (int64_t)B * H * W;
    {   // thread t gathers the patch of pixel t once (zero padding resolved here); everyone then reads it from LDS
        const int64_t p = (int64_t)blockIdx.x * NW_BLOCK_PIX + threadIdx.x;
        const int wq = (int)(p % W), hq = (int)((p / W) % H);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int hh = hq + t / 3 - 1, ww = wq + t % 3 - 1;
            const bool ok = p < M && hh >= 0 && hh < H && ww >= 0 && ww < W;
            const T* src = in + (p + (int64_t)(t / 3 - 1) * W + (t % 3 - 1)) * S;
#pragma unroll
            for (int j = 0; j < S; ++j) pat[threadIdx.x][t * S + j] = ok ? to_f32(src[j]) : 0.f;
        }
#pragma unroll
        for (int k = 9 * S; k < PS; ++k) pat[threadIdx.x][k] = 0.f;
    }
    __syncthreads();
    const int cg = (threadIdx.x & 31) * 8, pl = threadIdx.x >> 5;     // 32 groups of 8 channels x 8 pixel lanes
    if (cg >= cn) return;                                               // Cw % 8 == 0: a group is in or out as a whole
    f32x4 b0 = {0, 0, 0, 0}, b1 = {0, 0, 0, 0};
    if (bias) { b0 = load4(bias + c0 + cg); b1 = load4(bias + c0 + cg + 4); }
    for (int it = 0; it < NW_BLOCK_PIX / (8 * NW_PIX); ++it) {
        const int lp = (it * 8 + pl) * NW_PIX;                          // first of this thread's pixels inside the workgroup
        const int64_t p0 = (int64_t)blockIdx.x * NW_BLOCK_PIX + lp;
        f32x4 a0[NW_PIX], a1[NW_PIX];
#pragma unroll
        for (int q = 0; q < NW_PIX; ++q) { a0[q] = b0; a1[q] = b1; }
#pragma unroll
        for (int k4 = 0; k4 < PS / 4; ++k4) {
            f32x4 pv[NW_PIX];
#pragma unroll
            for (int q = 0; q < NW_PIX; ++q) pv[q] = load4(&pat[lp + q][4 * k4]);       // same address for the 32 channel groups: broadcast
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int kj = 4 * k4 + u;
                if (kj >= 9 * S) break;
                const f32x4 w0 = load4(&wl[kj][cg]), w1 = load4(&wl[kj][cg + 4]);
#pragma unroll
                for (int q = 0; q < NW_PIX; ++q) {
                    a0[q] += pv[q][u] * w0;
                    a1[q] += pv[q][u] * w1;
                }
            }
        }
#pragma unroll
        for (int q = 0; q < NW_PIX; ++q)
            if (p0 + q < M) {
                T* dst = out + (p0 + q) * Cw + c0 + cg;
                store4(dst, a0[q]);
                store4(dst + 4, a1[q]);
            }
    }
}

template <typename T, int S>
__global__ void __launch_bounds__(256)
conv3x3_narrow_out_kernel(const T* __restrict__ in, const T* __restrict__ w, const float* __restrict__ bias, T* __restrict__ out,
                          int B, int H, int W, int Cw) {
    extern __shared__ float wsm[];                   // [S][9][Cw] f32
    for (int i = threadIdx.x; i < S * 9 * Cw; i += 256) wsm[i] = to_f32(w[i]);
    __syncthreads();
    const int sub = threadIdx.x & 7;                 // 8 lanes share a pixel, each takes every 8th 8-channel chunk
    const int64_t M = (int64_t)B * H * W;
    const int64_t p = (int64_t)blockIdx.x * 32 + (threadIdx.x >> 3);
    const bool live = p < M;
    const int wq = live ? (int)(p % W) : 0, hq = live ? (int)((p / W) % H) : 0;
    float acc[S];
#pragma unroll
    for (int j = 0; j < S; ++j) acc[j] = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int hh = hq + t / 3 - 1, ww = wq + t % 3 - 1;
        if (!(live && hh >= 0 && hh < H && ww >= 0 && ww < W)) continue;
        const T* src = in + (p + (int64_t)(t / 3 - 1) * W + (t % 3 - 1)) * Cw;
        for (int c = sub * 8; c < Cw; c += 64) {
            const f32x4 x0 = load4(src + c), x1 = load4(src + c + 4);
#pragma unroll
            for (int j = 0; j < S; ++j) {
                const float* wr = wsm + ((int64_t)j * 9 + t) * Cw + c;
                const f32x4 w0 = load4(wr), w1 = load4(wr + 4);
                acc[j] += ((x0[0] * w0[0] + x0[1] * w0[1]) + (x0[2] * w0[2] + x0[3] * w0[3])) +
                          ((x1[0] * w1[0] + x1[1] * w1[1]) + (x1[2] * w1[2] + x1[3] * w1[3]));
            }
        }
    }
#pragma unroll
    for (int j = 0; j < S; ++j) {
        acc[j] += __shfl_xor(acc[j], 1, 64);
        acc[j] += __shfl_xor(acc[j], 2, 64);
        acc[j] += __shfl_xor(acc[j], 4, 64);
    }
    if (live && sub == 0) {
#pragma unroll
        for (int j = 0; j < S; ++j) out[p * S + j] = from_f32<T>(acc[j] + (bias ? bias[j] : 0.f));
    }
}

template <typename T, int S>
static void launch_narrow_out(const T* in, const T* w, const float* bias, T* out, int B, int H, int W, int Cw, int64_t M,
                              size_t lds, hipStream_t s) {
    (void)hipFuncSetAttribute((const void*)conv3x3_narrow_out_kernel<T, S>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    conv3x3_narrow_out_kernel<T, S><<<ceil_div(M, 32), 256, lds, s>>>(in, w, bias, out, B, H, W, Cw);
}

extern "C" int vaw_conv3x3_narrow(vaw_dtype dt, int mode, const void* in, const void* w, const float* bias, void* out, int B, int H,
                                  int W, int Cn, int Cw, vaw_stream stream) {
    VAW_CHECK_ARG(mode >= 0 && mode <= 2 && in && w && out && B > 0 && H > 0 && W > 0 && Cn > 0 && Cw > 0, "conv3x3_narrow: bad arguments");
    if (Cn > 4 || Cw % 8 || (mode == 2 && (int64_t)Cn * 9 * Cw * 4 > 96 * 1024)) return VAW_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const int64_t M = (int64_t)B * H * W;
    if (mode == 2) {
        const size_t lds = (size_t)Cn * 9 * Cw * 4;
#define NO_LAUNCH(Sv) BY_DTYPE(dt, (launch_narrow_out<T, Sv>((const T*)in, (const T*)w, bias, (T*)out, B, H, W, Cw, M, lds, s)))
        switch (Cn) {
            case 1: NO_LAUNCH(1); break;
            case 2: NO_LAUNCH(2); break;
            case 3: NO_LAUNCH(3); break;
            default: NO_LAUNCH(4); break;
        }
    } else {
        dim3 grid(ceil_div(M, NW_BLOCK_PIX), ceil_div(Cw, 256));
#define NI_LAUNCH(Sv) \
    BY_DTYPE(dt, (conv3x3_narrow_in_kernel<T, Sv><<<grid, 256, 0, s>>>((const T*)in, (const T*)w, bias, (T*)out, B, H, W, Cw, mode)))
        switch (Cn) {
            case 1: NI_LAUNCH(1); break;
            case 2: NI_LAUNCH(2); break;
            case 3: NI_LAUNCH(3); break;
            default: NI_LAUNCH(4); break;
        }
    }
    VAW_CHECK_LAUNCH("conv3x3_narrow");
    return VAW_OK;
}

// ---------------------------------------------------------------------------------------------
// resampling, concat, layout
// ---------------------------------------------------------------------------------------------
// mode 0: out[b,h,w,:] = mean of the 2x2 block of in (in is 2H x 2W)          (avg_pool2d; also nearest-upsample^T * 1/4 * 4)
// mode 1: out[b,h,w,:] = in[b,h/2,w/2,:] * s                                     (nearest x2; also avg_pool^T with s = 1/4)
template <typename T>
__global__ void resample2_kernel(const T* __restrict__ in, T* __restrict__ out, int B, int Ho, int Wo, int C, int mode, float s) {
    const int64_t total4 = (int64_t)B * Ho * Wo * C / 4;
    GRID_STRIDE(i, total4) {
        const int64_t e = 4 * i;
        const int c = (int)(e % C);
        int64_t r = e / C;
        const int w = (int)(r % Wo), h = (int)((r / Wo) % Ho), b = (int)(r / ((int64_t)Wo * Ho));
        f32x4 v;
        if (mode == 0) {
            const int Wi = 2 * Wo;
            const T* p = in + (((int64_t)b * 2 * Ho + 2 * h) * Wi + 2 * w) * C + c;
            v = ((load4(p) + load4(p + C)) + load4(p + (int64_t)Wi * C)) + load4(p + (int64_t)Wi * C + C);
            v *= s;
        } else {
            const int Wi = Wo / 2;
            v = load4(in + (((int64_t)b * (Ho / 2) + h / 2) * Wi + w / 2) * C + c) * s;
        }
        store4(out + e, v);
    }
}

// out[m, 0:Ca] = a[m,:], out[m, Ca:Ca+Cb] = b[m,:]   (split = the inverse; ADD accumulates into a/b instead of overwriting)
template <typename T, bool SPLIT>
__global__ void concat_kernel(T* __restrict__ a, T* __restrict__ b, T* __restrict__ cat, int64_t M, int Ca, int Cb) {
    const int C = Ca + Cb;
    const int64_t total4 = M * C / 4;
    GRID_STRIDE(i, total4) {
        const int64_t e = 4 * i;
        const int c = (int)(e % C);
        const int64_t m = e / C;
        T* side = c < Ca ? a + m * Ca + c : b + m * Cb + (c - Ca);
        if (SPLIT) store4(side, load4(cat + e));
        else store4(cat + e, load4(side));
    }
}

template <typename T>
__global__ void add_kernel(T* __restrict__ dst, const T* __restrict__ src, int64_t n) {
    GRID_STRIDE(i, n / 4) store4(dst + 4 * i, load4(dst + 4 * i) + load4(src + 4 * i));
}

// bf16 fast paths of the three data-movement kernels above (2x2 pool / nearest x2, channel concat / split, gradient add) on
// 16-byte accesses with U independent octets in flight per lane and non-temporal stores: they are pure HBM passes and ran at
// ~4 TB/s on 8-byte accesses with one load in flight and a 64-bit division per element.  Same arithmetic, same order.
#define MV_U 4
__device__ __forceinline__ void mv_unpack(gns_u32x4 v, float (&f)[8]) { gns_unpack(v, f); }
__device__ __forceinline__ gns_u32x4 mv_pack(const float (&f)[8]) {
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (bf16_t)f[j];
    return __builtin_bit_cast(gns_u32x4, v);
}
__global__ void __launch_bounds__(256)
add8_kernel(bf16_t* __restrict__ dst, const bf16_t* __restrict__ src, unsigned n8) {
    for (unsigned base = blockIdx.x * (256u * MV_U); base < n8; base += gridDim.x * (256u * MV_U)) {
        gns_u32x4 a[MV_U], b[MV_U];
#pragma unroll
        for (int u = 0; u < MV_U; ++u) {
            const unsigned i = base + u * 256u + threadIdx.x, ic = i < n8 ? i : n8 - 1;
            a[u] = *reinterpret_cast<const gns_u32x4*>(dst + 8ull * ic);
            b[u] = gns_ld_nt(src + 8ull * ic);
        }
#pragma unroll
        for (int u = 0; u < MV_U; ++u) {
            const unsigned i = base + u * 256u + threadIdx.x;
            float x[8], y[8];
            mv_unpack(a[u], x);
            mv_unpack(b[u], y);
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] += y[j];
            if (i < n8) *reinterpret_cast<gns_u32x4*>(dst + 8ull * i) = mv_pack(x);
        }
    }
}
template <bool SPLIT>
__global__ void __launch_bounds__(256)
concat8_kernel(bf16_t* __restrict__ a, bf16_t* __restrict__ b, bf16_t* __restrict__ cat, unsigned n8, unsigned Ca8, unsigned Cb8) {
    const unsigned C8 = Ca8 + Cb8;
    for (unsigned base = blockIdx.x * (256u * MV_U); base < n8; base += gridDim.x * (256u * MV_U)) {
        gns_u32x4 v[MV_U];
        bf16_t* side[MV_U];
#pragma unroll
        for (int u = 0; u < MV_U; ++u) {
            const unsigned i = base + u * 256u + threadIdx.x, ic = i < n8 ? i : n8 - 1;
            const unsigned m = ic / C8, c = ic - m * C8;
            side[u] = c < Ca8 ? a + 8ull * ((uint64_t)m * Ca8 + c) : b + 8ull * ((uint64_t)m * Cb8 + (c - Ca8));
            v[u] = SPLIT ? gns_ld_nt(cat + 8ull * ic) : gns_ld_nt(side[u]);
        }
#pragma unroll
        for (int u = 0; u < MV_U; ++u) {
            const unsigned i = base + u * 256u + threadIdx.x;
            if (i < n8) {
                if (SPLIT) *reinterpret_cast<gns_u32x4*>(side[u]) = v[u];
                else *reinterpret_cast<gns_u32x4*>(cat + 8ull * i) = v[u];
            }
        }
    }
}
template <int MODE>       // 0: 2x2 mean pool (out = Ho x Wo from 2Ho x 2Wo), 1: nearest x2 (out = Ho x Wo from Ho/2 x Wo/2); both times s
__global__ void __launch_bounds__(256)
resample8_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, unsigned n8, unsigned Ho, unsigned Wo, unsigned C8, float s) {
    for (unsigned base = blockIdx.x * (256u * MV_U); base < n8; base += gridDim.x * (256u * MV_U)) {
        gns_u32x4 v[MV_U][MODE == 0 ? 4 : 1];
#pragma unroll
        for (int u = 0; u < MV_U; ++u) {
            const unsigned i = base + u * 256u + threadIdx.x, ic = i < n8 ? i : n8 - 1;
            const unsigned r = ic / C8, c = ic - r * C8, w = r % Wo, hb = r / Wo, h = hb % Ho, b = hb / Ho;
            if (MODE == 0) {
                const unsigned Wi = 2 * Wo;
                const bf16_t* p = in + 8ull * ((((uint64_t)b * 2 * Ho + 2 * h) * Wi + 2 * w) * C8 + c);
                v[u][0] = gns_ld_nt(p);
                v[u][1] = gns_ld_nt(p + 8ull * C8);
                v[u][2] = gns_ld_nt(p + 8ull * Wi * C8);
                v[u][3] = gns_ld_nt(p + 8ull * Wi * C8 + 8ull * C8);
            } else {
                const unsigned Wi = Wo / 2;
                v[u][0] = gns_ld(in + 8ull * ((((uint64_t)b * (Ho / 2) + h / 2) * Wi + w / 2) * C8 + c));
            }
        }
#pragma unroll
        for (int u = 0; u < MV_U; ++u) {
            const unsigned i = base + u * 256u + threadIdx.x;
            float x[8];
            mv_unpack(v[u][0], x);
            if (MODE == 0) {
                float y[8], z[8], t[8];
                mv_unpack(v[u][1], y);
                mv_unpack(v[u][2], z);
                mv_unpack(v[u][MODE == 0 ? 3 : 0], t);
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = ((x[j] + y[j]) + z[j]) + t[j];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] *= s;
            if (i < n8) __builtin_nontemporal_store(mv_pack(x), reinterpret_cast<gns_u32x4*>(out + 8ull * i));
        }
    }
}
static inline int mv_grid(int64_t n8) {
    const int64_t g = (n8 + 256 * MV_U - 1) / (256 * MV_U);
    return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}
static bool mv_fast(vaw_dtype dt, int64_t n8) {
    static int on = -1;
    if (on < 0) {
        const char* v = getenv("VAW_MOVE8");
        on = v ? (atoi(v) != 0) : 1;
    }
    return on && dt == VAW_BF16 && n8 > 0 && n8 < ((int64_t)1 << 31);
}

// NCHW f32 <-> NHWC act dtype
template <typename T, bool TO_NHWC>
__global__ void layout_kernel(const float* __restrict__ nchw_in, float* __restrict__ nchw_out, const T* __restrict__ nhwc_in,
                              T* __restrict__ nhwc_out, int B, int C, int HW) {
    const int64_t total = (int64_t)B * C * HW;
    GRID_STRIDE(i, total) {
        const int p = (int)(i % HW);
        const int c = (int)((i / HW) % C);
        const int b = (int)(i / ((int64_t)HW * C));
        const int64_t j = ((int64_t)b * HW + p) * C + c;
        if (TO_NHWC) nhwc_out[j] = from_f32<T>(nchw_in[i]);
        else nchw_out[i] = to_f32(nhwc_in[j]);
    }
}

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
static inline int gn_chunks(int HW) { return (HW + GN_ROWS - 1) / GN_ROWS; }
extern "C" int64_t vaw_groupnorm_workspace_floats(int B, int HW, int C) {
    const int64_t streaming = (int64_t)4 * ceil_div(HW, GN_ROWS / 4) * B * C + (int64_t)4 * B * C + 2 * (int64_t)B * 64 + 64;   // the flat kernels' smallest chunks
    const int64_t coop = gnc_workspace_floats(B, HW, C);
    return streaming > coop ? streaming : coop;
}
// the cooperative single-read kernels (groupnorm_coop.h) unless VAW_GN_COOP=0 or the shape does not fit them
#if GNC_PROF
extern "C" int vaw_debug_gnc_prof(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(gnc_prof_buf), sizeof(unsigned long long) * n);
}
#endif
static bool gns_ok(vaw_dtype dt, int B, int HW, int C, int G) {
    static int on = -1;
    if (on < 0) {
        const char* v = getenv("VAW_GN_FLAT");
        on = v ? (atoi(v) != 0) : 1;
    }
    return on && dt == VAW_BF16 && C % 8 == 0 && C / 8 <= GNS_NT && C <= 2048 && C % G == 0 && gns_rows(B, HW) > 0 &&
           (int64_t)B * HW < (1 << 30) && (int64_t)HW * C < ((int64_t)1 << 31);
}
static int g_gn_coop = -1;      // -1: VAW_GN_COOP (default off: an experiment, see groupnorm_coop.h); 0 / 1: forced by vaw_debug_gn_coop
extern "C" void vaw_debug_gn_coop(int mode) { g_gn_coop = mode; }
static bool gnc_enabled() {
    static int env = -1;
    if (env < 0) {
        const char* v = getenv("VAW_GN_COOP");
        env = v ? (atoi(v) != 0) : 0;
    }
    return g_gn_coop >= 0 ? g_gn_coop != 0 : env != 0;
}
template <typename K>
static int gnc_grid_cached(K kernel, int block, int items) {
    static int per_block[GNC_NT / 64 + 2] = {0};        // resident workgroups on the chip for this kernel at this block size
    int& g = per_block[block / 64];
    if (!g) g = gnc_grid((const void*)kernel, block, 1 << 30);
    return g < items ? g : items;
}
#define GNC_FWD_CASE(S, F)                                                                                                       \
    if ((silu != 0) == S && (scale != nullptr) == F) {                                                                           \
        grid = gnc_grid_cached(gnc_fwd_kernel<GNC_NV_FWD, S, F>, block, gm.items);                                                           \
        if (grid >= gm.nch && gm.nch <= 64) {                                                                                      \
            if (hipMemsetAsync(part1, 0xff, sizeof(float) * B * gm.nch * 128, s) != hipSuccess) return -1;                       \
            gnc_fwd_kernel<GNC_NV_FWD, S, F><<<grid, block, 0, s>>>((const bf16_t*)x, gamma, beta, scale, shift, film_ld, (bf16_t*)y, mean,  \
                                                        rstd, HW, C, G, eps, gm, part1, counter);                                \
            return 1;                                                                                                            \
        }                                                                                                                        \
        return 0;                                                                                                                \
    }
// 1 = launched, 0 = not eligible (caller takes the streaming kernels), -1 = error
static int gnc_try_fwd(vaw_dtype dt, const void* x, const float* gamma, const float* beta, const float* scale, const float* shift,
                       int64_t film_ld, int silu, void* y, float* mean, float* rstd, int B, int HW, int C, int G, float eps,
                       float* workspace, hipStream_t s) {
    if (!gnc_enabled() || !gnc_shape_ok(dt, B, HW, C, G)) return 0;
    const GncGeom gm = gnc_geom(B, HW, C, GNC_NV_FWD);
    const int block = GNC_NT + 64;                       // 8 data waves + the sync wave
    float* part1 = workspace;
    unsigned* counter = reinterpret_cast<unsigned*>(workspace + (int64_t)B * gm.nch * 128);
    int grid = 0;
    GNC_FWD_CASE(true, true)
    GNC_FWD_CASE(true, false)
    GNC_FWD_CASE(false, true)
    GNC_FWD_CASE(false, false)
    return 0;
}
#define GNC_BWD_CASE(S, F, A)                                                                                                    \
    if ((silu != 0) == S && (scale != nullptr) == F && (dx_add != nullptr) == A) {                                               \
        grid = gnc_grid_cached(gnc_bwd_kernel<GNC_NV_BWD, S, F, A>, block, gm.items);                                                        \
        if (grid < gm.nch || gm.nch > 64) return 0;                                                                              \
        if (gm.nch > 1 && hipMemsetAsync(counter, 0, sizeof(unsigned) * B, s) != hipSuccess) return -1;                          \
        gnc_bwd_kernel<GNC_NV_BWD, S, F, A><<<grid, block, 0, s>>>((const bf16_t*)dout, (const bf16_t*)x, mean, rstd, gamma, beta, scale,    \
                                                       shift, film_ld, (const bf16_t*)dx_add, (bf16_t*)dx, HW, C, G, gm, part1,  \
                                                       part2, counter);                                                          \
        launched = true;                                                                                                         \
    }
static int gnc_try_bwd(vaw_dtype dt, const void* dout, const void* x, const float* mean, const float* rstd, const float* gamma,
                       const float* beta, const float* scale, const float* shift, int64_t film_ld, int silu, const void* dx_add,
                       void* dx, float* dgamma, float* dbeta, float grad_beta, float* dscale, float* dshift, int64_t dfilm_ld, int B,
                       int HW, int C, int G, float* workspace, hipStream_t s) {
    if (!gnc_enabled() || !gnc_shape_ok(dt, B, HW, C, G)) return 0;
    const GncGeom gm = gnc_geom(B, HW, C, GNC_NV_BWD);
    const int block = ((gm.nt + 63) / 64) * 64;
    const int64_t BC = (int64_t)B * C;
    float* part1 = workspace;
    float* part2 = part1 + (int64_t)B * gm.nch * 128;
    float* sums = part2 + (int64_t)B * gm.nch * 2 * C;
    unsigned* counter = reinterpret_cast<unsigned*>(sums + 4 * BC);
    int grid = 0;
    bool launched = false;
    GNC_BWD_CASE(true, true, true)
    GNC_BWD_CASE(true, true, false)
    GNC_BWD_CASE(true, false, true)
    GNC_BWD_CASE(true, false, false)
    GNC_BWD_CASE(false, true, true)
    GNC_BWD_CASE(false, true, false)
    GNC_BWD_CASE(false, false, true)
    GNC_BWD_CASE(false, false, false)
    if (!launched) return 0;
    gnc_bwd_fold_kernel<<<ceil_div(BC, 256), 256, 0, s>>>(part2, gm.nch, B, C, G, rstd, gamma, beta, scale, film_ld, sums);
    const int nb2 = (int)ceil_div(C, 16), nb3 = scale && dscale ? (int)ceil_div(BC, 256) : 0;
    gn_bwd_group_kernel<<<nb2 + nb3, 256, 0, s>>>(sums, 0, gamma, B, C, G, nullptr, nullptr, dgamma, dbeta, grad_beta,
                                                   scale ? dscale : nullptr, dshift, dfilm_ld, 0, nb2);
    return 1;
}

extern "C" int vaw_groupnorm_fwd(vaw_dtype dt, const void* x, const float* gamma, const float* beta, const float* scale,
                                 const float* shift, int64_t film_ld, int silu, void* y, float* mean, float* rstd, int B,
                                 int HW, int C, int G, float eps, float* workspace, vaw_stream stream) {
    VAW_CHECK_ARG(B > 0 && HW > 0 && C > 0 && G > 0 && G <= 64 && C % G == 0 && C % 4 == 0 && workspace && B < 65536,
                  "groupnorm_fwd: bad sizes (C=%d G=%d)", C, G);
    VAW_CHECK_ARG((scale == nullptr) == (shift == nullptr), "groupnorm_fwd: scale and shift go together");
    VAW_CHECK_ARG(!scale || film_ld % 4 == 0, "groupnorm_fwd: film_ld must be a multiple of 4");
    hipStream_t s = (hipStream_t)stream;
    const int coop = gnc_try_fwd(dt, x, gamma, beta, scale, shift, film_ld, silu, y, mean, rstd, B, HW, C, G, eps, workspace, s);
    VAW_CHECK_ARG(coop >= 0, "groupnorm_fwd: counter reset failed");
    if (coop == 1) {
        VAW_CHECK_LAUNCH("groupnorm_fwd");
        return VAW_OK;
    }
    if (gns_ok(dt, B, HW, C, G)) {
        const GnsGeom gm = gns_geom(C, gns_rows(B, HW));
        const int nch = ceil_div(HW, gm.rows);
        const bf16_t* xb = (const bf16_t*)x;
        bf16_t* yb = (bf16_t*)y;
        gns_fwd_sums_kernel<8><<<B * nch, GNS_NT, 0, s>>>(xb, HW, C, B, nch, gm, workspace);
        gn_group_stats_kernel<<<ceil_div(B * G, 128), 128, 0, s>>>(workspace, nch, B, C, G, HW, eps, mean, rstd);
#define GNS_APPLY(S, F)                                                                                                              \
    if ((silu != 0) == S && (scale != nullptr) == F)                                                                                 \
        gns_apply_kernel<4, S, F><<<B * nch, GNS_NT, 0, s>>>(xb, mean, rstd, gamma, beta, scale, shift, film_ld, yb, HW, C, G, nch, gm);
        GNS_APPLY(true, true)
        GNS_APPLY(true, false)
        GNS_APPLY(false, true)
        GNS_APPLY(false, false)
#undef GNS_APPLY
        VAW_CHECK_LAUNCH("groupnorm_fwd");
        return VAW_OK;
    }
    const int nch = gn_chunks(HW);
    dim3 grid(ceil_div(C, 64), B, nch);
    BY_DTYPE(dt, (gn_fwd_sums_kernel<T><<<grid, 256, 0, s>>>((const T*)x, HW, C, B, nch, workspace)));
    gn_group_stats_kernel<<<ceil_div(B * G, 128), 128, 0, s>>>(workspace, nch, B, C, G, HW, eps, mean, rstd);
    BY_DTYPE(dt, (gn_apply_kernel<T><<<grid, 256, 0, s>>>((const T*)x, mean, rstd, gamma, beta, scale, shift, film_ld, silu, (T*)y, HW, C, G)));
    VAW_CHECK_LAUNCH("groupnorm_fwd");
    return VAW_OK;
}

extern "C" int vaw_groupnorm_bwd(vaw_dtype dt, const void* dout, const void* x, const float* mean, const float* rstd,
                                 const float* gamma, const float* beta, const float* scale, const float* shift,
                                 int64_t film_ld, int silu, const void* dx_add, void* dx, float* dgamma, float* dbeta,
                                 float grad_beta, float* dscale, float* dshift, int64_t dfilm_ld, int B, int HW, int C, int G,
                                 float* workspace, vaw_stream stream) {
    VAW_CHECK_ARG(B > 0 && HW > 0 && C > 0 && G > 0 && G <= 64 && C % G == 0 && C % 4 == 0 && workspace && B < 65536,
                  "groupnorm_bwd: bad sizes");
    VAW_CHECK_ARG(!scale || (shift && dscale && dshift && film_ld % 4 == 0), "groupnorm_bwd: FiLM needs shift, dscale, dshift");
    hipStream_t s = (hipStream_t)stream;
    const int coop = gnc_try_bwd(dt, dout, x, mean, rstd, gamma, beta, scale, shift, film_ld, silu, dx_add, dx, dgamma, dbeta, grad_beta,
                                 dscale, dshift, dfilm_ld, B, HW, C, G, workspace, s);
    VAW_CHECK_ARG(coop >= 0, "groupnorm_bwd: counter reset failed");
    if (coop == 1) {
        VAW_CHECK_LAUNCH("groupnorm_bwd");
        return VAW_OK;
    }
    const bool flat = gns_ok(dt, B, HW, C, G);
    const GnsGeom gm = gns_geom(flat ? C : 8, flat ? gns_rows(B, HW) : GN_ROWS);
    const int nch = ceil_div(HW, gm.rows);
    const int64_t BC = (int64_t)B * C;
    float* part = workspace;
    float* sums = part + 4 * nch * BC;
    float* S1 = sums + 4 * BC;
    float* S2 = S1 + (int64_t)B * G;
    dim3 grid(ceil_div(C, 64), B, nch);
    const bf16_t *xb = (const bf16_t*)x, *db = (const bf16_t*)dout;
    if (flat) {
#define GNS_SUMS(S, F)                                                                                                             \
    if ((silu != 0) == S && (scale != nullptr) == F)                                                                               \
        gns_bwd_sums_kernel<4, S, F><<<B * nch, GNS_NT, 0, s>>>(db, xb, mean, rstd, gamma, beta, scale, shift, film_ld, HW, C, G, B, nch, gm, part);
        GNS_SUMS(true, true)
        GNS_SUMS(true, false)
        GNS_SUMS(false, true)
        GNS_SUMS(false, false)
#undef GNS_SUMS
    } else {
        BY_DTYPE(dt, (gn_bwd_sums_kernel<T><<<grid, 256, 0, s>>>((const T*)dout, (const T*)x, mean, rstd, gamma, beta, scale, shift, film_ld, silu, HW, C, G, B, nch, part)));
    }
    const int nb1 = (int)ceil_div(B * G, 256), nb2 = (int)ceil_div(C, 16), nb3 = scale && dscale ? (int)ceil_div(BC, 256) : 0;
    // (folding the chunks inside the group kernel -- nchunk > 0 -- saves the fold launch and was measured SLOWER: 5.4 + 4.9 us for the
    // pair against 15.6 us, its batch-strided lanes then walk nchunk x more dependent loads)
    gn_bwd_fold_kernel<<<ceil_div(4 * BC, 256), 256, 0, s>>>(part, nch, B, C, sums);
    gn_bwd_group_kernel<<<nb1 + nb2 + nb3, 256, 0, s>>>(sums, 0, gamma, B, C, G, S1, S2, dgamma, dbeta, grad_beta,
                                                         scale ? dscale : nullptr, dshift, dfilm_ld, nb1, nb2);
    if (flat) {
#define GNS_BAPPLY(S, F, A)                                                                                                        \
    if ((silu != 0) == S && (scale != nullptr) == F && (dx_add != nullptr) == A)                                                   \
        gns_bwd_apply_kernel<4, S, F, A><<<B * nch, GNS_NT, 0, s>>>(db, xb, mean, rstd, gamma, beta, scale, shift, film_ld, S1, S2, \
                                                                    (const bf16_t*)dx_add, (bf16_t*)dx, HW, C, G, nch, gm);
        GNS_BAPPLY(true, true, true)
        GNS_BAPPLY(true, true, false)
        GNS_BAPPLY(true, false, true)
        GNS_BAPPLY(true, false, false)
        GNS_BAPPLY(false, true, true)
        GNS_BAPPLY(false, true, false)
        GNS_BAPPLY(false, false, true)
        GNS_BAPPLY(false, false, false)
#undef GNS_BAPPLY
        VAW_CHECK_LAUNCH("groupnorm_bwd");
        return VAW_OK;
    }
    BY_DTYPE(dt, (gn_bwd_apply_kernel<T><<<grid, 256, 0, s>>>((const T*)dout, (const T*)x, mean, rstd, gamma, beta, scale, shift, film_ld, silu, S1, S2, (const T*)dx_add, (T*)dx, HW, C, G)));
    VAW_CHECK_LAUNCH("groupnorm_bwd");
    return VAW_OK;
}

extern "C" int vaw_im2col3x3(vaw_dtype dt, const void* x, void* col, int B, int H, int W, int C, vaw_stream stream) {
    VAW_CHECK_ARG(B > 0 && H > 0 && W > 0 && C > 0, "im2col3x3: bad sizes");
    hipStream_t s = (hipStream_t)stream;
    if (C % 4 == 0) {
        const int64_t n = (int64_t)B * H * W * 9 * (C / 4);
        BY_DTYPE(dt, (im2col3x3_kernel<T, 4><<<sgrid(n, 256), 256, 0, s>>>((const T*)x, (T*)col, B, H, W, C)));
    } else {
        const int64_t n = (int64_t)B * H * W * 9 * C;
        BY_DTYPE(dt, (im2col3x3_kernel<T, 1><<<sgrid(n, 256), 256, 0, s>>>((const T*)x, (T*)col, B, H, W, C)));
    }
    VAW_CHECK_LAUNCH("im2col3x3");
    return VAW_OK;
}

extern "C" int vaw_col2im3x3(vaw_dtype dt, const void* dcol, void* dx, int B, int H, int W, int C, vaw_stream stream) {
    VAW_CHECK_ARG(B > 0 && H > 0 && W > 0 && C > 0, "col2im3x3: bad sizes");
    hipStream_t s = (hipStream_t)stream;
    if (C % 4 == 0) {
        const int64_t n = (int64_t)B * H * W * (C / 4);
        BY_DTYPE(dt, (col2im3x3_kernel<T, 4><<<sgrid(n, 256), 256, 0, s>>>((const T*)dcol, (T*)dx, B, H, W, C)));
    } else {
        const int64_t n = (int64_t)B * H * W * C;
        BY_DTYPE(dt, (col2im3x3_kernel<T, 1><<<sgrid(n, 256), 256, 0, s>>>((const T*)dcol, (T*)dx, B, H, W, C)));
    }
    VAW_CHECK_LAUNCH("col2im3x3");
    return VAW_OK;
}

extern "C" int vaw_resample2(vaw_dtype dt, const void* in, void* out, int B, int Ho, int Wo, int C, int mode, float s_,
                             vaw_stream stream) {
    VAW_CHECK_ARG(B > 0 && Ho > 0 && Wo > 0 && C % 4 == 0 && (mode == 0 || (mode == 1 && Ho % 2 == 0 && Wo % 2 == 0)), "resample2: bad sizes");
    const int64_t n4 = (int64_t)B * Ho * Wo * C / 4;
    if (C % 8 == 0 && mv_fast(dt, n4 / 2) && ((uintptr_t)in | (uintptr_t)out) % 16 == 0) {
        const unsigned n8 = (unsigned)(n4 / 2);
        if (mode == 0) resample8_kernel<0><<<mv_grid(n8), 256, 0, (hipStream_t)stream>>>((const bf16_t*)in, (bf16_t*)out, n8, Ho, Wo, C / 8, s_);
        else resample8_kernel<1><<<mv_grid(n8), 256, 0, (hipStream_t)stream>>>((const bf16_t*)in, (bf16_t*)out, n8, Ho, Wo, C / 8, s_);
        VAW_CHECK_LAUNCH("resample2");
        return VAW_OK;
    }
    BY_DTYPE(dt, (resample2_kernel<T><<<sgrid(n4, 256), 256, 0, (hipStream_t)stream>>>((const T*)in, (T*)out, B, Ho, Wo, C, mode, s_)));
    VAW_CHECK_LAUNCH("resample2");
    return VAW_OK;
}

extern "C" int vaw_concat_channels(vaw_dtype dt, void* a, void* b, void* cat, int64_t M, int Ca, int Cb, int split,
                                   vaw_stream stream) {
    VAW_CHECK_ARG(M > 0 && Ca > 0 && Cb > 0 && Ca % 4 == 0 && Cb % 4 == 0, "concat_channels: channel counts must be multiples of 4");
    const int64_t n4 = M * (Ca + Cb) / 4;
    hipStream_t s = (hipStream_t)stream;
    if (Ca % 8 == 0 && Cb % 8 == 0 && mv_fast(dt, n4 / 2) && ((uintptr_t)a | (uintptr_t)b | (uintptr_t)cat) % 16 == 0) {
        const unsigned n8 = (unsigned)(n4 / 2);
        if (split) concat8_kernel<true><<<mv_grid(n8), 256, 0, s>>>((bf16_t*)a, (bf16_t*)b, (bf16_t*)cat, n8, Ca / 8, Cb / 8);
        else concat8_kernel<false><<<mv_grid(n8), 256, 0, s>>>((bf16_t*)a, (bf16_t*)b, (bf16_t*)cat, n8, Ca / 8, Cb / 8);
        VAW_CHECK_LAUNCH("concat_channels");
        return VAW_OK;
    }
    if (split) { BY_DTYPE(dt, (concat_kernel<T, true><<<sgrid(n4, 256), 256, 0, s>>>((T*)a, (T*)b, (T*)cat, M, Ca, Cb))); }
    else { BY_DTYPE(dt, (concat_kernel<T, false><<<sgrid(n4, 256), 256, 0, s>>>((T*)a, (T*)b, (T*)cat, M, Ca, Cb))); }
    VAW_CHECK_LAUNCH("concat_channels");
    return VAW_OK;
}

extern "C" int vaw_add_inplace(vaw_dtype dt, void* dst, const void* src, int64_t n, vaw_stream stream) {
    VAW_CHECK_ARG(n > 0 && n % 4 == 0, "add_inplace: n must be a positive multiple of 4");
    if (n % 8 == 0 && mv_fast(dt, n / 8) && ((uintptr_t)dst | (uintptr_t)src) % 16 == 0) {
        add8_kernel<<<mv_grid(n / 8), 256, 0, (hipStream_t)stream>>>((bf16_t*)dst, (const bf16_t*)src, (unsigned)(n / 8));
        VAW_CHECK_LAUNCH("add_inplace");
        return VAW_OK;
    }
    BY_DTYPE(dt, (add_kernel<T><<<sgrid(n / 4, 256), 256, 0, (hipStream_t)stream>>>((T*)dst, (const T*)src, n)));
    VAW_CHECK_LAUNCH("add_inplace");
    return VAW_OK;
}

// ---- ResBlock variants off the factory path (reference models/unet.py:81-140, 206-256) -------------------------------
// out = a * b elementwise (act dtype): nn.Dropout with a pre-scaled keep mask, forward and backward alike
template <typename T>
__global__ void mul_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out, int64_t n4) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x)
        store4(out + 4 * i, load4(a + 4 * i) * load4(b + 4 * i));
}
extern "C" int vaw_mul(vaw_dtype dt, const void* a, const void* b, void* out, int64_t n, vaw_stream stream) {
    VAW_CHECK_ARG(a && b && out && n > 0 && n % 4 == 0, "mul: n must be a positive multiple of 4");
    BY_DTYPE(dt, (mul_kernel<T><<<sgrid(n / 4, 256), 256, 0, (hipStream_t)stream>>>((const T*)a, (const T*)b, (T*)out, n / 4)));
    VAW_CHECK_LAUNCH("mul");
    return VAW_OK;
}

// mode 0: out[b,i,j,:] = in[b,2i,2j,:]  (the even pixels: a stride-2 conv is the stride-1 conv sampled there)
// mode 1: out[b,i,j,:] = (i, j both even) ? in[b,i/2,j/2,:] : 0   (its transpose); out is [B,Ho,Wo,C] in both modes
template <typename T>
__global__ void subsample2_kernel(const T* __restrict__ in, T* __restrict__ out, int B, int Ho, int Wo, int C, int mode) {
    const int64_t n4 = (int64_t)B * Ho * Wo * C / 4;
    const int c4n = C / 4;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % c4n);
        const int64_t pix = i / c4n;
        const int w = (int)(pix % Wo), h = (int)((pix / Wo) % Ho);
        const int64_t b = pix / ((int64_t)Wo * Ho);
        f32x4 v = {0, 0, 0, 0};
        if (mode == 0) v = load4(in + ((b * 2 * Ho + 2 * h) * (2 * Wo) + 2 * w) * C + 4 * c4);
        else if (!(h & 1) && !(w & 1)) v = load4(in + ((b * (Ho / 2) + h / 2) * (Wo / 2) + w / 2) * C + 4 * c4);
        store4(out + 4 * i, v);
    }
}
extern "C" int vaw_subsample2(vaw_dtype dt, const void* in, void* out, int B, int Ho, int Wo, int C, int mode, vaw_stream stream) {
    VAW_CHECK_ARG(in && out && B > 0 && Ho > 0 && Wo > 0 && C % 4 == 0 && (mode == 0 || (mode == 1 && Ho % 2 == 0 && Wo % 2 == 0)),
                  "subsample2: bad sizes");
    const int64_t n4 = (int64_t)B * Ho * Wo * C / 4;
    BY_DTYPE(dt, (subsample2_kernel<T><<<sgrid(n4, 256), 256, 0, (hipStream_t)stream>>>((const T*)in, (T*)out, B, Ho, Wo, C, mode)));
    VAW_CHECK_LAUNCH("subsample2");
    return VAW_OK;
}

// h[b,p,:] += e[b,:]  (use_scale_shift_norm=False: h + emb_out, reference :250-252); e is f32 with row stride ld
template <typename T>
__global__ void rowvec_add_kernel(T* __restrict__ h, const float* __restrict__ e, int64_t ld, int HW, int C, int64_t n4) {
    const int c4n = C / 4;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % c4n);
        const int64_t b = i / ((int64_t)c4n * HW);
        store4(h + 4 * i, load4(h + 4 * i) + load4(e + b * ld + 4 * c4));
    }
}
extern "C" int vaw_rowvec_add(vaw_dtype dt, void* h, const float* e, int64_t ld, int B, int HW, int C, vaw_stream stream) {
    VAW_CHECK_ARG(h && e && B > 0 && HW > 0 && C % 4 == 0 && ld % 4 == 0 && ((uintptr_t)e & 15) == 0, "rowvec_add: bad sizes");
    const int64_t n4 = (int64_t)B * HW * C / 4;
    BY_DTYPE(dt, (rowvec_add_kernel<T><<<sgrid(n4, 256), 256, 0, (hipStream_t)stream>>>((T*)h, e, ld, HW, C, n4)));
    VAW_CHECK_LAUNCH("rowvec_add");
    return VAW_OK;
}
// de[b,c] = beta * de[b,c] + sum_p dh[b,p,c]: one workgroup per (sample, 64-channel slab), fixed-order tree
template <typename T>
__global__ void rowvec_sum_kernel(const T* __restrict__ dh, float* __restrict__ de, int64_t ld, int HW, int C, float beta) {
    __shared__ float part[4][64];
    const int b = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), r = threadIdx.x >> 6;
    float acc = 0.f;
    if (c < C)
        for (int p = r; p < HW; p += 4) acc += to_f32(dh[((int64_t)b * HW + p) * C + c]);
    part[r][threadIdx.x & 63] = acc;
    __syncthreads();
    if (r == 0 && c < C) {
        const float t = ((part[0][threadIdx.x] + part[1][threadIdx.x]) + part[2][threadIdx.x]) + part[3][threadIdx.x];
        float* o = de + (int64_t)b * ld + c;
        *o = (beta != 0.f ? beta * *o : 0.f) + t;
    }
}
extern "C" int vaw_rowvec_sum(vaw_dtype dt, const void* dh, float* de, int64_t ld, int B, int HW, int C, float beta, vaw_stream stream) {
    VAW_CHECK_ARG(dh && de && B > 0 && HW > 0 && C > 0 && B < 65536, "rowvec_sum: bad sizes");
    dim3 grid(ceil_div(C, 64), B);
    BY_DTYPE(dt, (rowvec_sum_kernel<T><<<grid, 256, 0, (hipStream_t)stream>>>((const T*)dh, de, ld, HW, C, beta)));
    VAW_CHECK_LAUNCH("rowvec_sum");
    return VAW_OK;
}

extern "C" int vaw_nchw_to_nhwc(vaw_dtype dt, const float* nchw, void* nhwc, int B, int C, int HW, vaw_stream stream) {
    VAW_CHECK_ARG(B > 0 && C > 0 && HW > 0, "nchw_to_nhwc: bad sizes");
    const int64_t n = (int64_t)B * C * HW;
    BY_DTYPE(dt, (layout_kernel<T, true><<<sgrid(n, 256), 256, 0, (hipStream_t)stream>>>(nchw, nullptr, nullptr, (T*)nhwc, B, C, HW)));
    VAW_CHECK_LAUNCH("nchw_to_nhwc");
    return VAW_OK;
}
extern "C" int vaw_nhwc_to_nchw(vaw_dtype dt, const void* nhwc, float* nchw, int B, int C, int HW, vaw_stream stream) {
    VAW_CHECK_ARG(B > 0 && C > 0 && HW > 0, "nhwc_to_nchw: bad sizes");
    const int64_t n = (int64_t)B * C * HW;
    BY_DTYPE(dt, (layout_kernel<T, false><<<sgrid(n, 256), 256, 0, (hipStream_t)stream>>>(nullptr, nchw, (const T*)nhwc, nullptr, B, C, HW)));
    VAW_CHECK_LAUNCH("nhwc_to_nchw");
    return VAW_OK;
}
