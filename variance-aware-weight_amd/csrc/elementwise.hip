// HBM-bound kernels of the diffusion objective and the optimizer: q_sample, weighted MSE,
// sinusoidal embedding, SiLU, embedding gather/scatter, patchify/unpatchify, AdamW+EMA.
// All are coalesced 16-byte streams (float4 / bf16x4), grid-strided, capped at 8 blocks per CU.
#include "common.h"

static inline int stream_grid(int64_t work_items, int block) {
    int64_t g = (work_items + block - 1) / block;
    const int64_t cap = 256 * 8;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

// ---------------------------------------------------------------------------------------------
// out[b,:] = ca[b]*x[b,:] + cb[b]*y[b,:]; coefficients either given per row or gathered from tables.
// ---------------------------------------------------------------------------------------------
template <bool GATHER>
__global__ void mix_rows_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ ca,
                                const float* __restrict__ cb, const int64_t* __restrict__ t, int T,
                                float* __restrict__ out, int64_t n) {
    const int b = blockIdx.y;
    float a, c;
    if (GATHER) {
        const int64_t tt = t[b];
        const bool ok = tt >= 0 && tt < T;
        a = ok ? ca[tt] : __builtin_nanf("");
        c = ok ? cb[tt] : __builtin_nanf("");
    } else {
        a = ca[b];
        c = cb[b];
    }
    const float* xr = x + (int64_t)b * n;
    const float* yr = y + (int64_t)b * n;
    float* orow = out + (int64_t)b * n;
    const int64_t n4 = ((n & 3) == 0) ? n / 4 : 0;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        f32x4 xv = load4(xr + 4 * i), yv = load4(yr + 4 * i);
        store4(orow + 4 * i, a * xv + c * yv);
    }
    for (int64_t i = n4 * 4 + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        orow[i] = a * xr[i] + c * yr[i];
}

extern "C" int vaw_qsample_fwd(const float* x0, const float* noise, const int64_t* t, const float* tab_a,
                               const float* tab_s, int num_timesteps, float* x_t, int B, int64_t per_sample,
                               vaw_stream stream) {
    VAW_CHECK_ARG(B > 0 && per_sample > 0 && num_timesteps > 0, "qsample: bad sizes B=%d n=%ld", B, (long)per_sample);
    int gx = stream_grid(per_sample / 4 + 1, 256);
    dim3 grid(gx > 64 ? 64 : gx, B);
    mix_rows_kernel<true><<<grid, 256, 0, (hipStream_t)stream>>>(x0, noise, tab_a, tab_s, t, num_timesteps, x_t, per_sample);
    VAW_CHECK_LAUNCH("qsample");
    return VAW_OK;
}

extern "C" int vaw_mix_rows(const float* x, const float* y, const float* ca, const float* cb, float* out, int B,
                            int64_t per_sample, vaw_stream stream) {
    VAW_CHECK_ARG(B > 0 && per_sample > 0, "mix_rows: bad sizes");
    int gx = stream_grid(per_sample / 4 + 1, 256);
    dim3 grid(gx > 64 ? 64 : gx, B);
    mix_rows_kernel<false><<<grid, 256, 0, (hipStream_t)stream>>>(x, y, ca, cb, nullptr, 0, out, per_sample);
    VAW_CHECK_LAUNCH("mix_rows");
    return VAW_OK;
}

// ---------------------------------------------------------------------------------------------
// Weighted MSE: one block per sample; wave shuffles + one LDS hop for the reduction.
// ---------------------------------------------------------------------------------------------
__global__ void wmse_fwd_kernel(const float* __restrict__ o, const float* __restrict__ x0, const float* __restrict__ nz,
                                const float* __restrict__ ca, const float* __restrict__ cb, const float* __restrict__ w,
                                float* __restrict__ mse, int64_t n) {
    __shared__ float scratch[16];
    const int b = blockIdx.x;
    const float a = ca[b], c = cb[b];
    const float* orow = o + (int64_t)b * n;
    const float* xr = x0 + (int64_t)b * n;
    const float* nr = nz + (int64_t)b * n;
    float acc = 0.f;
    const int64_t n4 = ((n & 3) == 0) ? n / 4 : 0;
    for (int64_t i = threadIdx.x; i < n4; i += blockDim.x) {
        f32x4 d = a * load4(xr + 4 * i) + c * load4(nr + 4 * i) - load4(orow + 4 * i);
        acc += d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3];
    }
    for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) {
        float d = a * xr[i] + c * nr[i] - orow[i];
        acc += d * d;
    }
    float tot = block_sum(acc, scratch);
    if (threadIdx.x == 0) mse[b] = w[b] * (tot / (float)n);
}

__global__ void wmse_bwd_kernel(const float* __restrict__ o, const float* __restrict__ x0, const float* __restrict__ nz,
                                const float* __restrict__ ca, const float* __restrict__ cb, const float* __restrict__ w,
                                const float* __restrict__ gmse, float* __restrict__ dout, int64_t n) {
    const int b = blockIdx.y;
    const float a = ca[b], c = cb[b];
    const float k = gmse[b] * w[b] * 2.f / (float)n;
    const float* orow = o + (int64_t)b * n;
    const float* xr = x0 + (int64_t)b * n;
    const float* nr = nz + (int64_t)b * n;
    float* dr = dout + (int64_t)b * n;
    const int64_t n4 = ((n & 3) == 0) ? n / 4 : 0;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x)
        store4(dr + 4 * i, k * (load4(orow + 4 * i) - a * load4(xr + 4 * i) - c * load4(nr + 4 * i)));
    for (int64_t i = n4 * 4 + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dr[i] = k * (orow[i] - a * xr[i] - c * nr[i]);
}

extern "C" int vaw_wmse_fwd(const float* model_out, const float* x0, const float* noise, const float* ca,
                            const float* cb, const float* w, float* mse, int B, int64_t per_sample,
                            vaw_stream stream) {
    VAW_CHECK_ARG(B > 0 && per_sample > 0, "wmse_fwd: bad sizes");
    wmse_fwd_kernel<<<B, 1024, 0, (hipStream_t)stream>>>(model_out, x0, noise, ca, cb, w, mse, per_sample);
    VAW_CHECK_LAUNCH("wmse_fwd");
    return VAW_OK;
}

extern "C" int vaw_wmse_bwd(const float* model_out, const float* x0, const float* noise, const float* ca,
                            const float* cb, const float* w, const float* gmse, float* dout, int B,
                            int64_t per_sample, vaw_stream stream) {
    VAW_CHECK_ARG(B > 0 && per_sample > 0, "wmse_bwd: bad sizes");
    int gx = stream_grid(per_sample / 4 + 1, 256);
    dim3 grid(gx > 64 ? 64 : gx, B);
    wmse_bwd_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(model_out, x0, noise, ca, cb, w, gmse, dout, per_sample);
    VAW_CHECK_LAUNCH("wmse_bwd");
    return VAW_OK;
}

// ---------------------------------------------------------------------------------------------
// Variational-bound term (learned variance / KL losses): one pass over (x0, x_t, model mean, model var values) per
// sample, bits per dim.  coef[b][8] = {c1, c2, plv, lv_aux, pa, pb, is_t0, -}: posterior mean coefficients, the
// clipped posterior log variance (true log variance and the lower end of the learned range), log(beta_t) (upper end
// of the range) or the fixed model log variance, pred_xstart = pa*x_t + pb*mean_out, and the t == 0 flag.
//   mean_mode 0: model mean = c1*pred_xstart + c2*x_t      1: model mean = mean_out (PREVIOUS_X)
//   var_mode  0: log variance = lv_aux (fixed)   1: = var values (LEARNED)   2: interpolated (LEARNED_RANGE)
// ---------------------------------------------------------------------------------------------
#define VB_NCOEF 8
struct VbElem {
    float val;      // KL or decoder NLL of this element, nats
    float d_lv;     // d val / d model log variance
    float d_mean;   // d val / d model mean
};
__device__ __forceinline__ float vb_cdf(float z, float& dcdf) {
    const float k = 0.7978845608028654f;                       // sqrt(2/pi)
    const float th = tanhf(k * (z + 0.044715f * (z * z * z)));
    dcdf = 0.5f * (1.f - th * th) * k * (1.f + 3.f * 0.044715f * z * z);
    return 0.5f * (1.f + th);
}
__device__ __forceinline__ VbElem vb_elem(float x0, float true_mean, float true_lv, float mean, float lv, bool t0) {
    VbElem r;
    if (!t0) {                                                  // normal_kl (tools/losses.py:12-39)
        const float e2 = expf(-lv), d = true_mean - mean, ratio = expf(true_lv - lv);
        r.val = 0.5f * (-1.0f + lv - true_lv + ratio + (d * d) * e2);
        r.d_lv = 0.5f * (1.f - ratio - (d * d) * e2);
        r.d_mean = -d * e2;
        return r;
    }
    // -discretized_gaussian_log_likelihood(x0; mean, 0.5*lv) (tools/losses.py:50-76)
    const float cx = x0 - mean, inv = expf(-(0.5f * lv));
    const float pin = inv * (cx + 1.0f / 255.0f), mnn = inv * (cx - 1.0f / 255.0f);
    float dp, dm;
    const float cp = vb_cdf(pin, dp), cm = vb_cdf(mnn, dm);
    float lp, g_pin = 0.f, g_min = 0.f;                         // d log_prob / d plus_in, / d min_in
    if (x0 < -0.999f) {
        lp = logf(fmaxf(cp, 1e-12f));
        if (cp >= 1e-12f) g_pin = dp / cp;
    } else if (x0 > 0.999f) {
        const float q = 1.f - cm;
        lp = logf(fmaxf(q, 1e-12f));
        if (q >= 1e-12f) g_min = -dm / q;
    } else {
        const float dl = cp - cm;
        lp = logf(fmaxf(dl, 1e-12f));
        if (dl >= 1e-12f) { g_pin = dp / dl; g_min = -dm / dl; }
    }
    r.val = -lp;
    r.d_lv = 0.5f * (g_pin * pin + g_min * mnn);               // d plus_in / d lv = -plus_in / 2
    r.d_mean = inv * (g_pin + g_min);                           // d plus_in / d mean = -inv_stdv
    return r;
}
__device__ __forceinline__ void vb_model(const float* c, int mean_mode, int var_mode, float xt, float m, float v, float& mean,
                                         float& lv) {
    mean = mean_mode == 1 ? m : c[0] * (c[4] * xt + c[5] * m) + c[1] * xt;
    if (var_mode == 1) lv = v;
    else if (var_mode == 2) { const float frac = (v + 1.f) / 2.f; lv = frac * c[3] + (1.f - frac) * c[2]; }
    else lv = c[3];
}

__global__ void vb_fwd_kernel(const float* __restrict__ mean_out, const float* __restrict__ var_out, const float* __restrict__ x0,
                              const float* __restrict__ xt, const float* __restrict__ coef, int mean_mode, int var_mode,
                              float scale, float* __restrict__ vb, int64_t n) {
    __shared__ float scratch[16];
    const int b = blockIdx.x;
    float c[VB_NCOEF];
#pragma unroll
    for (int i = 0; i < VB_NCOEF; ++i) c[i] = coef[b * VB_NCOEF + i];
    const bool t0 = c[6] != 0.f;
    const int64_t base = (int64_t)b * n;
    float acc = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
        const float x = x0[base + i], z = xt[base + i];
        float mean, lv;
        vb_model(c, mean_mode, var_mode, z, mean_out[base + i], var_out ? var_out[base + i] : 0.f, mean, lv);
        acc += vb_elem(x, c[0] * x + c[1] * z, c[2], mean, lv, t0).val;
    }
    const float tot = block_sum(acc, scratch);
    if (threadIdx.x == 0) vb[b] = scale * ((tot / (float)n) / 0.6931471805599453f);
}

__global__ void vb_bwd_kernel(const float* __restrict__ mean_out, const float* __restrict__ var_out, const float* __restrict__ x0,
                              const float* __restrict__ xt, const float* __restrict__ coef, int mean_mode, int var_mode,
                              float scale, const float* __restrict__ gvb, float* __restrict__ d_mean, float* __restrict__ d_var,
                              int64_t n) {
    const int b = blockIdx.y;
    float c[VB_NCOEF];
#pragma unroll
    for (int i = 0; i < VB_NCOEF; ++i) c[i] = coef[b * VB_NCOEF + i];
    const bool t0 = c[6] != 0.f;
    const float g = gvb[b] * scale / ((float)n * 0.6931471805599453f);
    const float dlv_dv = var_mode == 2 ? 0.5f * (c[3] - c[2]) : 1.f;
    const float dmean_dm = mean_mode == 1 ? 1.f : c[0] * c[5];
    const int64_t base = (int64_t)b * n;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float x = x0[base + i], z = xt[base + i];
        float mean, lv;
        vb_model(c, mean_mode, var_mode, z, mean_out[base + i], var_out ? var_out[base + i] : 0.f, mean, lv);
        const VbElem e = vb_elem(x, c[0] * x + c[1] * z, c[2], mean, lv, t0);
        if (d_var) d_var[base + i] = g * e.d_lv * dlv_dv;
        if (d_mean) d_mean[base + i] = g * e.d_mean * dmean_dm;
    }
}

extern "C" int vaw_vb_fwd(const float* mean_out, const float* var_out, const float* x0, const float* x_t, const float* coef,
                          int mean_mode, int var_mode, float scale, float* vb, int B, int64_t per_sample, vaw_stream stream) {
    VAW_CHECK_ARG(B > 0 && per_sample > 0 && mean_out && x0 && x_t && coef && vb, "vb_fwd: bad arguments");
    VAW_CHECK_ARG((mean_mode == 0 || mean_mode == 1) && var_mode >= 0 && var_mode <= 2 && (var_mode == 0 || var_out),
                  "vb_fwd: bad modes (learned variance needs var_out)");
    vb_fwd_kernel<<<B, 1024, 0, (hipStream_t)stream>>>(mean_out, var_out, x0, x_t, coef, mean_mode, var_mode, scale, vb, per_sample);
    VAW_CHECK_LAUNCH("vb_fwd");
    return VAW_OK;
}

extern "C" int vaw_vb_bwd(const float* mean_out, const float* var_out, const float* x0, const float* x_t, const float* coef,
                          int mean_mode, int var_mode, float scale, const float* gvb, float* d_mean, float* d_var, int B,
                          int64_t per_sample, vaw_stream stream) {
    VAW_CHECK_ARG(B > 0 && per_sample > 0 && mean_out && x0 && x_t && coef && gvb && (d_mean || d_var), "vb_bwd: bad arguments");
    VAW_CHECK_ARG((mean_mode == 0 || mean_mode == 1) && var_mode >= 0 && var_mode <= 2 && (var_mode == 0 || var_out) &&
                      (var_mode != 0 || !d_var),
                  "vb_bwd: bad modes");
    int gx = stream_grid(per_sample, 256);
    dim3 grid(gx > 64 ? 64 : gx, B);
    vb_bwd_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(mean_out, var_out, x0, x_t, coef, mean_mode, var_mode, scale, gvb, d_mean,
                                                         d_var, per_sample);
    VAW_CHECK_LAUNCH("vb_bwd");
    return VAW_OK;
}

// ---------------------------------------------------------------------------------------------
// One reverse-process step (sampling side): p_mean_variance + p_sample / ddim_sample fused.
// coef[b][16] = {pa, pb, c1, c2, plv, lv_aux, ra, rm1, sqrt_abp, s1, abp, is_t0, s2, -, -, -}:
//   pred_xstart = pa*x + pb*mean_out (clamped to [-1,1] if clip), model mean = c1*pred + c2*x (or mean_out itself),
//   eps = (ra*x - pred)/rm1, sigma = (eta*s1)*s2, ddim mean = pred*sqrt_abp + sqrt(1 - abp - sigma^2)*eps.
// kind 0: no sample (p_mean_variance only)   1: ancestral p_sample   2: ddim_sample.  Outputs may be NULL.
// ---------------------------------------------------------------------------------------------
#define SS_NCOEF 16
__global__ void sample_step_kernel(const float* __restrict__ mean_out, const float* __restrict__ var_out, const float* __restrict__ x,
                                   const float* __restrict__ noise, const float* __restrict__ coef, int kind, int mean_mode,
                                   int var_mode, int clip, float eta, float* __restrict__ sample, float* __restrict__ pred_out,
                                   float* __restrict__ mean_o, float* __restrict__ logvar_o, int64_t n) {
    const int b = blockIdx.y;
    float c[SS_NCOEF];
#pragma unroll
    for (int i = 0; i < SS_NCOEF; ++i) c[i] = coef[b * SS_NCOEF + i];
    const float mask = c[11] != 0.f ? 0.f : 1.f;
    const float sigma = (eta * c[9]) * c[12];
    const float ddim_c = sqrtf(1.f - c[10] - sigma * sigma);
    const int64_t base = (int64_t)b * n;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float xv = x[base + i], m = mean_out[base + i];
        float pred = c[0] * xv + c[1] * m;
        if (clip) pred = fminf(fmaxf(pred, -1.f), 1.f);
        float lv;
        if (var_mode == 1) lv = var_out[base + i];
        else if (var_mode == 2) { const float frac = (var_out[base + i] + 1.f) / 2.f; lv = frac * c[5] + (1.f - frac) * c[4]; }
        else lv = c[5];
        const float mean = mean_mode == 1 ? m : c[2] * pred + c[3] * xv;
        if (pred_out) pred_out[base + i] = pred;
        if (mean_o) mean_o[base + i] = mean;
        if (logvar_o) logvar_o[base + i] = lv;
        if (kind == 1) {
            sample[base + i] = mean + (mask * expf(0.5f * lv)) * noise[base + i];
        } else if (kind == 2) {
            const float eps = (c[6] * xv - pred) / c[7];
            const float mp = pred * c[8] + ddim_c * eps;
            sample[base + i] = mp + (mask * sigma) * noise[base + i];
        }
    }
}

extern "C" int vaw_sample_step(int kind, const float* mean_out, const float* var_out, const float* x, const float* noise,
                               const float* coef, int mean_mode, int var_mode, int clip_denoised, float eta, float* sample,
                               float* pred_xstart, float* mean, float* log_variance, int B, int64_t per_sample,
                               vaw_stream stream) {
    VAW_CHECK_ARG(B > 0 && per_sample > 0 && mean_out && x && coef && kind >= 0 && kind <= 2, "sample_step: bad arguments");
    VAW_CHECK_ARG((mean_mode == 0 || mean_mode == 1) && var_mode >= 0 && var_mode <= 2 && (var_mode == 0 || var_out),
                  "sample_step: bad modes (learned variance needs var_out)");
    VAW_CHECK_ARG(kind == 0 || (sample && noise), "sample_step: kind 1/2 need noise and sample");
    int gx = stream_grid(per_sample, 256);
    dim3 grid(gx > 64 ? 64 : gx, B);
    sample_step_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(mean_out, var_out, x, noise, coef, kind, mean_mode, var_mode,
                                                             clip_denoised, eta, sample, pred_xstart, mean, log_variance, per_sample);
    VAW_CHECK_LAUNCH("sample_step");
    return VAW_OK;
}

// ---------------------------------------------------------------------------------------------
// Small conditioning-path kernels ([B, D]-sized)
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void timestep_embedding_kernel(const float* __restrict__ t, T* __restrict__ out, int B, int dim,
                                          float neg_log_period) {
    const int half = dim / 2;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < (int64_t)B * dim;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(idx / dim), j = (int)(idx % dim);
        float v = 0.f;
        if (j < 2 * half) {
            const int i = j < half ? j : j - half;
            // same op order as the reference: exp(-ln(P) * i / half) in f32, then t * f
            const float f = expf(neg_log_period * (float)i / (float)half);
            const float ang = t[b] * f;
            v = j < half ? cosf(ang) : sinf(ang);
        }
        out[idx] = from_f32<T>(v);
    }
}

extern "C" int vaw_timestep_embedding(vaw_dtype dt, const float* t, void* out, int B, int dim, float max_period,
                                      vaw_stream stream) {
    VAW_CHECK_ARG(B > 0 && dim > 0, "timestep_embedding: bad sizes");
    const float nl = -logf(max_period);
    int grid = stream_grid((int64_t)B * dim, 256);
    if (dt == VAW_F32)
        timestep_embedding_kernel<float><<<grid, 256, 0, (hipStream_t)stream>>>(t, (float*)out, B, dim, nl);
    else
        timestep_embedding_kernel<bf16_t><<<grid, 256, 0, (hipStream_t)stream>>>(t, (bf16_t*)out, B, dim, nl);
    VAW_CHECK_LAUNCH("timestep_embedding");
    return VAW_OK;
}

template <typename T>
__global__ void silu_fwd_kernel(const float* __restrict__ x, T* __restrict__ out, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = from_f32<T>(silu_f(x[i]));
}
__global__ void silu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dx[i] = dy[i] * silu_grad_f(x[i]);
}
extern "C" int vaw_silu_fwd(vaw_dtype dt, const float* x, void* out, int64_t n, vaw_stream stream) {
    VAW_CHECK_ARG(n > 0, "silu_fwd: n<=0");
    int grid = stream_grid(n, 256);
    if (dt == VAW_F32) silu_fwd_kernel<float><<<grid, 256, 0, (hipStream_t)stream>>>(x, (float*)out, n);
    else silu_fwd_kernel<bf16_t><<<grid, 256, 0, (hipStream_t)stream>>>(x, (bf16_t*)out, n);
    VAW_CHECK_LAUNCH("silu_fwd");
    return VAW_OK;
}
extern "C" int vaw_silu_bwd(const float* x, const float* dy, float* dx, int64_t n, vaw_stream stream) {
    VAW_CHECK_ARG(n > 0, "silu_bwd: n<=0");
    silu_bwd_kernel<<<stream_grid(n, 256), 256, 0, (hipStream_t)stream>>>(x, dy, dx, n);
    VAW_CHECK_LAUNCH("silu_bwd");
    return VAW_OK;
}

__global__ void add_embedding_kernel(const float* __restrict__ a, const float* __restrict__ table,
                                     const int64_t* __restrict__ idx, float* __restrict__ out, int B, int D, int rows) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < (int64_t)B * D; i += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / D), d = (int)(i % D);
        const int64_t r = idx[b];
        out[i] = a[i] + ((r >= 0 && r < rows) ? table[r * D + d] : __builtin_nanf(""));
    }
}
// dtable[r,:] = beta*dtable[r,:] + sum over {b : idx[b]==r} dc[b,:], b ascending.  One workgroup per table row: the index vector
// goes through LDS in chunks of 1024 (every thread reads the same word: a broadcast), a thread owns columns t, t + 256, ... and adds
// the matching rows in batch order.  Deterministic, no pre-zeroing pass over the table.  (Round 3 had one thread per table ELEMENT
// scanning the whole index vector from global memory: 53 us for a 1001 x 768 table at batch 256; this form takes ~5.)
__global__ void __launch_bounds__(256)
embedding_bwd_kernel(const float* __restrict__ dc, const int64_t* __restrict__ idx, float* __restrict__ dtable, int B, int D,
                     int rows, float beta) {
    __shared__ int s_idx[1024];
    const int r = blockIdx.x;
    constexpr int MAXC = 8;                                   // columns per thread held in registers (D <= 2048); wider tables loop
    for (int d0 = 0; d0 < D; d0 += 256 * MAXC) {
        float acc[MAXC];
#pragma unroll
        for (int j = 0; j < MAXC; ++j) acc[j] = 0.f;
        for (int b0 = 0; b0 < B; b0 += 1024) {
            const int nb = B - b0 < 1024 ? B - b0 : 1024;
            __syncthreads();
            for (int i = threadIdx.x; i < nb; i += 256) s_idx[i] = (int)idx[b0 + i];
            __syncthreads();
            for (int i = 0; i < nb; ++i) {
                if (s_idx[i] != r) continue;                  // uniform branch
                const float* src = dc + (int64_t)(b0 + i) * D + d0;
#pragma unroll
                for (int j = 0; j < MAXC; ++j) {
                    const int d = threadIdx.x + 256 * j;
                    if (d0 + d < D) acc[j] += src[d];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < MAXC; ++j) {
            const int d = d0 + threadIdx.x + 256 * j;
            if (d < D) {
                float* o = dtable + (int64_t)r * D + d;
                *o = (beta != 0.f ? beta * *o : 0.f) + acc[j];
            }
        }
    }
}
extern "C" int vaw_add_embedding(const float* a, const float* table, const int64_t* idx, float* out, int B, int D,
                                 int num_rows, vaw_stream stream) {
    VAW_CHECK_ARG(B > 0 && D > 0 && num_rows > 0, "add_embedding: bad sizes");
    add_embedding_kernel<<<stream_grid((int64_t)B * D, 256), 256, 0, (hipStream_t)stream>>>(a, table, idx, out, B, D, num_rows);
    VAW_CHECK_LAUNCH("add_embedding");
    return VAW_OK;
}
extern "C" int vaw_embedding_bwd(const float* dc, const int64_t* idx, float* dtable, int B, int D, int num_rows,
                                 float beta, vaw_stream stream) {
    VAW_CHECK_ARG(B > 0 && D > 0 && num_rows > 0, "embedding_bwd: bad sizes");
    embedding_bwd_kernel<<<num_rows, 256, 0, (hipStream_t)stream>>>(dc, idx, dtable, B, D, num_rows, beta);
    VAW_CHECK_LAUNCH("embedding_bwd");
    return VAW_OK;
}

// ---------------------------------------------------------------------------------------------
// patchify / unpatchify: index shuffles between NCHW images and token rows.
//   patch-embed token column  k = (c*p + i)*p + j      (Conv2d weight [D, C, p, p] flattened)
//   final-layer token column  k = (i*p + j)*C + c      (einsum nhwpqc->nchpwq, dit.py:253-255)
// One thread per image element; the image side is always the coalesced one.
// ---------------------------------------------------------------------------------------------
template <typename T, bool CONV_ORDER, bool TO_TOKENS>
__global__ void patch_shuffle_kernel(const float* __restrict__ img_in, float* __restrict__ img_out,
                                     const float* __restrict__ tok_in_f32, T* __restrict__ tok_out, int B, int C, int H,
                                     int W, int p) {
    const int hp = H / p, wp = W / p;
    const int64_t total = (int64_t)B * C * H * W;
    const int K = C * p * p;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        int x = (int)(e % W);
        int64_t r = e / W;
        int y = (int)(r % H);
        r /= H;
        int c = (int)(r % C);
        int b = (int)(r / C);
        const int ty = y / p, i = y % p, tx = x / p, j = x % p;
        const int64_t row = ((int64_t)b * hp + ty) * wp + tx;
        const int col = CONV_ORDER ? (c * p + i) * p + j : (i * p + j) * C + c;
        if (TO_TOKENS) tok_out[row * K + col] = from_f32<T>(img_in[e]);
        else img_out[e] = tok_in_f32[row * K + col];
    }
}

extern "C" int vaw_patchify(vaw_dtype dt, const float* img, void* tok, int B, int C, int H, int W, int p,
                            vaw_stream stream) {
    VAW_CHECK_ARG(B > 0 && C > 0 && p > 0 && H % p == 0 && W % p == 0, "patchify: bad sizes");
    int grid = stream_grid((int64_t)B * C * H * W, 256);
    if (dt == VAW_F32)
        patch_shuffle_kernel<float, true, true><<<grid, 256, 0, (hipStream_t)stream>>>(img, nullptr, nullptr, (float*)tok, B, C, H, W, p);
    else
        patch_shuffle_kernel<bf16_t, true, true><<<grid, 256, 0, (hipStream_t)stream>>>(img, nullptr, nullptr, (bf16_t*)tok, B, C, H, W, p);
    VAW_CHECK_LAUNCH("patchify");
    return VAW_OK;
}
extern "C" int vaw_patchify_bwd(const float* dtok, float* dimg, int B, int C, int H, int W, int p, vaw_stream stream) {
    VAW_CHECK_ARG(B > 0 && C > 0 && p > 0 && H % p == 0 && W % p == 0, "patchify_bwd: bad sizes");
    int grid = stream_grid((int64_t)B * C * H * W, 256);
    patch_shuffle_kernel<float, true, false><<<grid, 256, 0, (hipStream_t)stream>>>(nullptr, dimg, dtok, nullptr, B, C, H, W, p);
    VAW_CHECK_LAUNCH("patchify_bwd");
    return VAW_OK;
}
extern "C" int vaw_unpatchify(vaw_dtype dt, const float* tok, float* img, int B, int C, int H, int W, int p,
                              vaw_stream stream) {
    (void)dt;
    VAW_CHECK_ARG(B > 0 && C > 0 && p > 0 && H % p == 0 && W % p == 0, "unpatchify: bad sizes");
    int grid = stream_grid((int64_t)B * C * H * W, 256);
    patch_shuffle_kernel<float, false, false><<<grid, 256, 0, (hipStream_t)stream>>>(nullptr, img, tok, nullptr, B, C, H, W, p);
    VAW_CHECK_LAUNCH("unpatchify");
    return VAW_OK;
}
extern "C" int vaw_unpatchify_bwd(vaw_dtype dt, const float* dimg, void* dtok, int B, int C, int H, int W, int p,
                                  vaw_stream stream) {
    VAW_CHECK_ARG(B > 0 && C > 0 && p > 0 && H % p == 0 && W % p == 0, "unpatchify_bwd: bad sizes");
    int grid = stream_grid((int64_t)B * C * H * W, 256);
    if (dt == VAW_F32)
        patch_shuffle_kernel<float, false, true><<<grid, 256, 0, (hipStream_t)stream>>>(dimg, nullptr, nullptr, (float*)dtok, B, C, H, W, p);
    else
        patch_shuffle_kernel<bf16_t, false, true><<<grid, 256, 0, (hipStream_t)stream>>>(dimg, nullptr, nullptr, (bf16_t*)dtok, B, C, H, W, p);
    VAW_CHECK_LAUNCH("unpatchify_bwd");
    return VAW_OK;
}

// ---------------------------------------------------------------------------------------------
// Optimizer: sum of squares, fused AdamW + EMA + bf16 shadow, EMA alone, cast.
// 36 B/param algorithmic traffic for the fused pass (p,g,m,v,ema read; p,m,v,ema written) + 2 B shadow.
// ---------------------------------------------------------------------------------------------
#define SUMSQ_MAX_BLOCKS 2048
__global__ void sumsq_partial_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ partial) {
    __shared__ float scratch[16];
    float acc = 0.f;
    const int64_t n4 = n / 4;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        f32x4 v = load4(g + 4 * i);
        acc += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    for (int64_t i = n4 * 4 + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        acc += g[i] * g[i];
    float tot = block_sum(acc, scratch);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}
__global__ void sumsq_final_kernel(const float* __restrict__ partial, int nblk, float* __restrict__ out, int accumulate) {
    __shared__ float scratch[16];
    float acc = 0.f;
    for (int i = threadIdx.x; i < nblk; i += blockDim.x) acc += partial[i];
    float tot = block_sum(acc, scratch);
    if (threadIdx.x == 0) out[0] = (accumulate ? out[0] : 0.f) + tot;
}
extern "C" int64_t vaw_sumsq_workspace_floats(void) { return SUMSQ_MAX_BLOCKS; }
extern "C" int vaw_sumsq(const float* g, int64_t n, float* sumsq_out, int accumulate, float* workspace,
                         vaw_stream stream) {
    VAW_CHECK_ARG(n > 0 && ((uintptr_t)g & 15) == 0 && workspace, "sumsq: n<=0, unaligned or no workspace");
    const int grid = stream_grid(n / 4 + 1, 256);
    sumsq_partial_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(g, n, workspace);
    sumsq_final_kernel<<<1, 256, 0, (hipStream_t)stream>>>(workspace, grid, sumsq_out, accumulate);
    VAW_CHECK_LAUNCH("sumsq");
    return VAW_OK;
}

struct AdamArgs {
    float lr, b1, b2, eps, wd, bc1, bc2_sqrt, ema_decay, clip;
    int zero_grad;
};

__device__ __forceinline__ void adam_one(float& p, float& g, float& m, float& v, float& e, bool has_ema, float gs,
                                         const AdamArgs& a) {
    const float gg = g * gs;
    p = p * (1.f - a.lr * a.wd);
    m = m + (gg - m) * (1.f - a.b1);
    v = v * a.b2 + (1.f - a.b2) * gg * gg;
    const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
    p = p - (a.lr / a.bc1) * (m / denom);
    if (has_ema) e = e * a.ema_decay + p * (1.f - a.ema_decay);
}

// Each workgroup walks contiguous tiles of ADAM_U x 256 float4 per stream (16 KiB of p, g, m, v and ema each): every lane has
// ADAM_U independent 16-byte loads of all five streams in flight before the first use (20 loads per lane instead of 5), a stream's
// accesses stay inside one DRAM page run per tile instead of hopping 8 MiB between iterations, and every byte is touched once per
// step (38 B per parameter, 5 GB for DiT-B: nothing of it is in a cache when the next step comes round) -> non-temporal loads and
// stores.  Round 3's one-float4-per-iteration grid-stride loop ran at 5.0 TB/s of these bytes.
#define ADAM_U 4
__device__ __forceinline__ f32x4 nt_load4(const float* p) { return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p)); }
__global__ void __launch_bounds__(256)
adamw_ema_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                 float* __restrict__ ema, bf16_t* __restrict__ shadow, int64_t n, const float* __restrict__ sumsq, AdamArgs a,
                 const float* __restrict__ hyper) {
    if (hyper) {          // step-dependent scalars from device memory: the launch can then sit in a replayed hipGraph
        a.lr = hyper[0];
        a.bc1 = hyper[1];
        a.bc2_sqrt = sqrtf(hyper[2]);
    }
    float gs = 1.f;
    if (a.clip > 0.f) {
        const float coef = a.clip / (sqrtf(*sumsq) + 1e-6f);
        gs = coef < 1.f ? coef : 1.f;
    }
    const int64_t n4 = n / 4;
    const bool has_ema = ema != nullptr;
    constexpr int64_t TILE = (int64_t)ADAM_U * 256;                  // float4 per tile
    const int64_t n_tiles = n4 / TILE;
    for (int64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const int64_t i0 = t * TILE + threadIdx.x;
        f32x4 pv[ADAM_U], gv[ADAM_U], mv[ADAM_U], vv[ADAM_U], ev[ADAM_U];
#pragma unroll
        for (int u = 0; u < ADAM_U; ++u) {
            const int64_t i = 4 * (i0 + 256 * u);
            pv[u] = nt_load4(p + i); gv[u] = nt_load4(g + i); mv[u] = nt_load4(m + i); vv[u] = nt_load4(v + i);
            ev[u] = has_ema ? nt_load4(ema + i) : f32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (int u = 0; u < ADAM_U; ++u) {
            const int64_t i = 4 * (i0 + 256 * u);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float pj = pv[u][j], gj = gv[u][j], mj = mv[u][j], vj = vv[u][j], ej = ev[u][j];
                adam_one(pj, gj, mj, vj, ej, has_ema, gs, a);
                pv[u][j] = pj; mv[u][j] = mj; vv[u][j] = vj; ev[u][j] = ej;
            }
            __builtin_nontemporal_store(pv[u], reinterpret_cast<f32x4*>(p + i));
            __builtin_nontemporal_store(mv[u], reinterpret_cast<f32x4*>(m + i));
            __builtin_nontemporal_store(vv[u], reinterpret_cast<f32x4*>(v + i));
            if (has_ema) __builtin_nontemporal_store(ev[u], reinterpret_cast<f32x4*>(ema + i));
            if (shadow) store4(shadow + i, pv[u]);          // default policy: the next forward reads the shadow weights first
            if (a.zero_grad) store4(g + i, f32x4{0, 0, 0, 0});
        }
    }
    // the last, partial tile (and n % 4 elements), one element group per thread
    for (int64_t i = n_tiles * TILE + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        f32x4 pv = load4(p + 4 * i), gv = load4(g + 4 * i), mv = load4(m + 4 * i), vv = load4(v + 4 * i);
        f32x4 ev = has_ema ? load4(ema + 4 * i) : f32x4{0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float pj = pv[j], gj = gv[j], mj = mv[j], vj = vv[j], ej = ev[j];
            adam_one(pj, gj, mj, vj, ej, has_ema, gs, a);
            pv[j] = pj; mv[j] = mj; vv[j] = vj; ev[j] = ej;
        }
        store4(p + 4 * i, pv);
        store4(m + 4 * i, mv);
        store4(v + 4 * i, vv);
        if (has_ema) store4(ema + 4 * i, ev);
        if (shadow) store4(shadow + 4 * i, pv);
        if (a.zero_grad) store4(g + 4 * i, f32x4{0, 0, 0, 0});
    }
    for (int64_t i = n4 * 4 + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float pv = p[i], gv = g[i], mv = m[i], vv = v[i], ev = has_ema ? ema[i] : 0.f;
        adam_one(pv, gv, mv, vv, ev, has_ema, gs, a);
        p[i] = pv; m[i] = mv; v[i] = vv;
        if (has_ema) ema[i] = ev;
        if (shadow) shadow[i] = (bf16_t)pv;
        if (a.zero_grad) g[i] = 0.f;
    }
}
static inline int adam_grid(int64_t n) {
    const int64_t tiles = n / 4 / (ADAM_U * 256);
    const int64_t cap = 256 * 8;
    return (int)(tiles < 1 ? 1 : (tiles > cap ? cap : tiles));
}

extern "C" int vaw_adamw_ema_step(float* p, float* g, float* m, float* v, float* ema, void* shadow_bf16, int64_t n,
                                  float lr, float beta1, float beta2, float eps, float weight_decay, float bc1,
                                  float bc2, float ema_decay, const float* sumsq, float clip_max_norm, int zero_grad,
                                  vaw_stream stream) {
    VAW_CHECK_ARG(n > 0, "adamw: n<=0");
    VAW_CHECK_ARG((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v | (uintptr_t)ema) & 15) == 0 &&
                      ((uintptr_t)shadow_bf16 & 7) == 0, "adamw: buffers must be 16-byte aligned");
    VAW_CHECK_ARG(clip_max_norm <= 0.f || sumsq != nullptr, "adamw: clip needs sumsq");
    AdamArgs a{lr, beta1, beta2, eps, weight_decay, bc1, sqrtf(bc2), ema_decay, clip_max_norm, zero_grad};
    adamw_ema_kernel<<<adam_grid(n), 256, 0, (hipStream_t)stream>>>(p, g, m, v, ema, (bf16_t*)shadow_bf16, n, sumsq, a, nullptr);
    VAW_CHECK_LAUNCH("adamw_ema");
    return VAW_OK;
}

extern "C" int vaw_adamw_ema_step_dev(float* p, float* g, float* m, float* v, float* ema, void* shadow_bf16, int64_t n,
                                      const float* hyper, float beta1, float beta2, float eps, float weight_decay,
                                      float ema_decay, const float* sumsq, float clip_max_norm, int zero_grad,
                                      vaw_stream stream) {
    VAW_CHECK_ARG(n > 0 && hyper, "adamw_dev: n<=0 or no hyper buffer");
    VAW_CHECK_ARG((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v | (uintptr_t)ema) & 15) == 0 &&
                      ((uintptr_t)shadow_bf16 & 7) == 0, "adamw_dev: buffers must be 16-byte aligned");
    VAW_CHECK_ARG(clip_max_norm <= 0.f || sumsq != nullptr, "adamw_dev: clip needs sumsq");
    AdamArgs a{0.f, beta1, beta2, eps, weight_decay, 1.f, 1.f, ema_decay, clip_max_norm, zero_grad};
    adamw_ema_kernel<<<adam_grid(n), 256, 0, (hipStream_t)stream>>>(p, g, m, v, ema, (bf16_t*)shadow_bf16, n, sumsq, a, hyper);
    VAW_CHECK_LAUNCH("adamw_ema_dev");
    return VAW_OK;
}

__global__ void ema_kernel(float* __restrict__ ema, const float* __restrict__ src, int64_t n, float d) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        ema[i] = ema[i] * d + src[i] * (1.f - d);
}
extern "C" int vaw_ema_update(float* ema, const float* src, int64_t n, float decay, vaw_stream stream) {
    VAW_CHECK_ARG(n > 0, "ema: n<=0");
    ema_kernel<<<stream_grid(n, 256), 256, 0, (hipStream_t)stream>>>(ema, src, n, decay);
    VAW_CHECK_LAUNCH("ema");
    return VAW_OK;
}

__global__ void cast_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int64_t n) {
    const int64_t n4 = n / 4;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x)
        store4(dst + 4 * i, load4(src + 4 * i));
    for (int64_t i = n4 * 4 + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = (bf16_t)src[i];
}
extern "C" int vaw_cast_bf16(const float* src, void* dst, int64_t n, vaw_stream stream) {
    VAW_CHECK_ARG(n > 0 && ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 7) == 0, "cast_bf16: n<=0 or unaligned");
    cast_bf16_kernel<<<stream_grid(n / 4 + 1, 256), 256, 0, (hipStream_t)stream>>>(src, (bf16_t*)dst, n);
    VAW_CHECK_LAUNCH("cast_bf16");
    return VAW_OK;
}

// dst[i] = scale * float(src[i]): gradient buckets that travelled over xGMI in bf16 come back into the f32 gradient
// buffer (scale = 1/world when the collective summed instead of averaging).
__global__ void uncast_bf16_kernel(const bf16_t* __restrict__ src, float* __restrict__ dst, int64_t n, float scale) {
    const int64_t n8 = n / 8;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        const bf16x8 v = reinterpret_cast<const bf16x8*>(src)[i];
        f32x4 a = {(float)v[0], (float)v[1], (float)v[2], (float)v[3]}, b = {(float)v[4], (float)v[5], (float)v[6], (float)v[7]};
        reinterpret_cast<f32x4*>(dst)[2 * i] = a * scale;
        reinterpret_cast<f32x4*>(dst)[2 * i + 1] = b * scale;
    }
    for (int64_t i = n8 * 8 + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = scale * (float)src[i];
}
extern "C" int vaw_uncast_bf16(const void* src, float* dst, int64_t n, float scale, vaw_stream stream) {
    VAW_CHECK_ARG(n > 0 && ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0, "uncast_bf16: n<=0 or unaligned");
    uncast_bf16_kernel<<<stream_grid(n / 8 + 1, 256), 256, 0, (hipStream_t)stream>>>((const bf16_t*)src, dst, n, scale);
    VAW_CHECK_LAUNCH("uncast_bf16");
    return VAW_OK;
}
