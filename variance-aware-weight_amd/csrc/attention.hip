// Multi-head attention softmax(scale * q k^T) v with arbitrary (batch, head, token, channel) strides, so the
// token-major DiT layout (timm Attention) and the channel-major UNet layout (QKVAttention) share kernels.
//
// "rowwise" family (this file, round 1): one wavefront owns one query row (forward, dQ) or one key row
// (dK/dV); lanes run over keys for the score pass and over channels for the value pass, softmax
// statistics are wave shuffles, probabilities cross from the key-indexed to the channel-indexed pass
// through a per-wave LDS row.  Exact f32 arithmetic on f32 or bf16 storage: this is the parity-mode
// kernel and the fallback for shapes the MFMA kernel does not take.  Backward recomputes P from the saved
// row log-sum-exp (no T x T tensor is ever written), and is split in a query-major and a key-major kernel
// so that no gradient needs atomics: results are bitwise reproducible.
#include "common.h"

#define ATT_MAX_TILES 16   // T <= 64*16

struct AttnDev {
    int B, H, T, hd;
    int64_t q_sb, q_sh, q_st, q_sd;
    int64_t o_sb, o_sh, o_st, o_sd;
    float scale;
};

template <typename T>
__device__ __forceinline__ float ldg(const T* p, int64_t off) { return to_f32(p[off]); }

// grid: (ceil(T/4), B*H); block 256 = 4 waves = 4 query rows
template <typename T>
__global__ void __launch_bounds__(256)
attn_fwd_rowwise(AttnDev a, const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v, T* __restrict__ o,
                 float* __restrict__ lse) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // per wave: [hd] q row + [T] probabilities
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int bh = blockIdx.y, b = bh / a.H, h = bh % a.H;
    const int i = blockIdx.x * 4 + wid;
    float* qs = lds + wid * (a.hd + a.T);
    float* ps = qs + a.hd;
    const int64_t base = b * a.q_sb + h * a.q_sh;
    if (i < a.T)
        for (int d = lane; d < a.hd; d += 64) qs[d] = ldg(q, base + i * a.q_st + d * a.q_sd);
    __syncthreads();
    if (i >= a.T) return;
    float s[ATT_MAX_TILES];
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < ATT_MAX_TILES; ++t) {
        const int j = t * 64 + lane;
        s[t] = -INFINITY;
        if (t * 64 < a.T && j < a.T) {
            float acc = 0.f;
            const int64_t kb = base + j * a.q_st;
            for (int d = 0; d < a.hd; ++d) acc += qs[d] * ldg(k, kb + d * a.q_sd);
            s[t] = acc * a.scale;
            mx = fmaxf(mx, s[t]);
        }
    }
    mx = wave_max(mx);
    float l = 0.f;
#pragma unroll
    for (int t = 0; t < ATT_MAX_TILES; ++t) {
        const int j = t * 64 + lane;
        if (t * 64 < a.T && j < a.T) {
            s[t] = __expf(s[t] - mx);
            l += s[t];
        }
    }
    l = wave_sum(l);
    const float inv = 1.f / l;
#pragma unroll
    for (int t = 0; t < ATT_MAX_TILES; ++t) {
        const int j = t * 64 + lane;
        if (t * 64 < a.T && j < a.T) ps[j] = s[t] * inv;
    }
    if (lane == 0) lse[(int64_t)bh * a.T + i] = mx + __logf(l);
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const int64_t ob = b * a.o_sb + h * a.o_sh + i * a.o_st;
    for (int d = lane; d < a.hd; d += 64) {
        float acc = 0.f;
        const int64_t vb = base + d * a.q_sd;
        for (int j = 0; j < a.T; ++j) acc += ps[j] * ldg(v, vb + j * a.q_st);
        o[ob + d * a.o_sd] = from_f32<T>(acc);
    }
}

// Query-major backward: delta_i, dS row, dQ row.
template <typename T>
__global__ void __launch_bounds__(256)
attn_bwd_q_rowwise(AttnDev a, const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v,
                   const T* __restrict__ o, const T* __restrict__ d_o, const float* __restrict__ lse,
                   float* __restrict__ delta, T* __restrict__ dq) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // per wave: [hd] q, [hd] dO, [T] dS
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int bh = blockIdx.y, b = bh / a.H, h = bh % a.H;
    const int i = blockIdx.x * 4 + wid;
    float* qs = lds + wid * (2 * a.hd + a.T);
    float* dos = qs + a.hd;
    float* ds = dos + a.hd;
    const int64_t base = b * a.q_sb + h * a.q_sh;
    const int64_t ob = b * a.o_sb + h * a.o_sh + (int64_t)i * a.o_st;
    float dl = 0.f;
    if (i < a.T) {
        for (int d = lane; d < a.hd; d += 64) {
            qs[d] = ldg(q, base + i * a.q_st + d * a.q_sd);
            const float g = ldg(d_o, ob + d * a.o_sd);
            dos[d] = g;
            dl += g * ldg(o, ob + d * a.o_sd);
        }
    }
    __syncthreads();
    if (i >= a.T) return;
    dl = wave_sum(dl);
    const float li = lse[(int64_t)bh * a.T + i];
    if (lane == 0) delta[(int64_t)bh * a.T + i] = dl;
    for (int j = lane; j < a.T; j += 64) {
        float sc = 0.f, dp = 0.f;
        const int64_t kb = base + j * a.q_st;
        for (int d = 0; d < a.hd; ++d) {
            sc += qs[d] * ldg(k, kb + d * a.q_sd);
            dp += dos[d] * ldg(v, kb + d * a.q_sd);
        }
        const float p = __expf(sc * a.scale - li);
        ds[j] = p * (dp - dl);
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    for (int d = lane; d < a.hd; d += 64) {
        float acc = 0.f;
        const int64_t kb = base + d * a.q_sd;
        for (int j = 0; j < a.T; ++j) acc += ds[j] * ldg(k, kb + j * a.q_st);
        dq[base + i * a.q_st + d * a.q_sd] = from_f32<T>(acc * a.scale);
    }
}

// Key-major backward: dK row, dV row.
template <typename T>
__global__ void __launch_bounds__(256)
attn_bwd_kv_rowwise(AttnDev a, const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v,
                    const T* __restrict__ d_o, const float* __restrict__ lse, const float* __restrict__ delta,
                    T* __restrict__ dk, T* __restrict__ dv) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // per wave: [hd] k, [hd] v, [T] P, [T] dS
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int bh = blockIdx.y, b = bh / a.H, h = bh % a.H;
    const int j = blockIdx.x * 4 + wid;
    float* ks = lds + wid * (2 * a.hd + 2 * a.T);
    float* vs = ks + a.hd;
    float* ps = vs + a.hd;
    float* ds = ps + a.T;
    const int64_t base = b * a.q_sb + h * a.q_sh;
    const int64_t obase = b * a.o_sb + h * a.o_sh;
    if (j < a.T)
        for (int d = lane; d < a.hd; d += 64) {
            ks[d] = ldg(k, base + j * a.q_st + d * a.q_sd);
            vs[d] = ldg(v, base + j * a.q_st + d * a.q_sd);
        }
    __syncthreads();
    if (j >= a.T) return;
    for (int i = lane; i < a.T; i += 64) {
        float sc = 0.f, dp = 0.f;
        const int64_t qb = base + i * a.q_st, ob = obase + i * a.o_st;
        for (int d = 0; d < a.hd; ++d) {
            sc += ks[d] * ldg(q, qb + d * a.q_sd);
            dp += vs[d] * ldg(d_o, ob + d * a.o_sd);
        }
        const float p = __expf(sc * a.scale - lse[(int64_t)bh * a.T + i]);
        ps[i] = p;
        ds[i] = p * (dp - delta[(int64_t)bh * a.T + i]);
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    for (int d = lane; d < a.hd; d += 64) {
        float av = 0.f, ak = 0.f;
        for (int i = 0; i < a.T; ++i) {
            av += ps[i] * ldg(d_o, obase + i * a.o_st + d * a.o_sd);
            ak += ds[i] * ldg(q, base + i * a.q_st + d * a.q_sd);
        }
        dv[base + j * a.q_st + d * a.q_sd] = from_f32<T>(av);
        dk[base + j * a.q_st + d * a.q_sd] = from_f32<T>(ak * a.scale);
    }
}

// bf16 MFMA path (attention_mfma.hip)
bool vaw_attn_mfma_ok(vaw_dtype dt, const vaw_attn_desc* d, const void* q, const void* k, const void* v, const void* o);
int vaw_attn_fwd_mfma(const vaw_attn_desc* d, const void* q, const void* k, const void* v, void* o, float* lse, hipStream_t s);
int vaw_attn_bwd_mfma(const vaw_attn_desc* d, const void* q, const void* k, const void* v, const void* o, const void* d_o,
                      const float* lse, float* delta, void* dq, void* dk, void* dv, hipStream_t s, float* cs_part = nullptr,
                      int64_t* cs_rows_out = nullptr);
static int g_force_rowwise = 0;
extern "C" void vaw_debug_force_rowwise_attention(int on) { g_force_rowwise = on; }

static int check_desc(const vaw_attn_desc* d, const char* who) {
    VAW_CHECK_ARG(d && d->B > 0 && d->H > 0 && d->T > 0 && d->hd > 0, "%s: bad descriptor", who);
    VAW_CHECK_ARG(d->T <= 64 * ATT_MAX_TILES, "%s: T=%d exceeds %d", who, d->T, 64 * ATT_MAX_TILES);
    VAW_CHECK_ARG((int64_t)d->B * d->H < 65536, "%s: B*H=%ld exceeds grid.y", who, (long)d->B * d->H);
    return VAW_OK;
}
static AttnDev to_dev(const vaw_attn_desc* d) {
    AttnDev a{d->B, d->H, d->T, d->hd, d->q_sb, d->q_sh, d->q_st, d->q_sd, d->o_sb, d->o_sh, d->o_st, d->o_sd, d->scale};
    return a;
}

extern "C" int vaw_attn_fwd(vaw_dtype dt, const vaw_attn_desc* d, const void* q, const void* k, const void* v, void* o,
                            float* lse, vaw_stream stream) {
    int rc = check_desc(d, "attn_fwd");
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (!g_force_rowwise && vaw_attn_mfma_ok(dt, d, q, k, v, o)) return vaw_attn_fwd_mfma(d, q, k, v, o, lse, s);
    AttnDev a = to_dev(d);
    dim3 grid(ceil_div(a.T, 4), a.B * a.H);
    const size_t lds = 4 * (size_t)(a.hd + a.T) * sizeof(float);
    if (dt == VAW_F32)
        attn_fwd_rowwise<float><<<grid, 256, lds, s>>>(a, (const float*)q, (const float*)k, (const float*)v, (float*)o, lse);
    else
        attn_fwd_rowwise<bf16_t><<<grid, 256, lds, s>>>(a, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)o, lse);
    VAW_CHECK_LAUNCH("attn_fwd");
    return VAW_OK;
}

extern "C" int vaw_attn_bwd(vaw_dtype dt, const vaw_attn_desc* d, const void* q, const void* k, const void* v,
                            const void* o, const void* d_o, const float* lse, float* delta, void* dq, void* dk, void* dv,
                            vaw_stream stream) {
    int rc = check_desc(d, "attn_bwd");
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (!g_force_rowwise && vaw_attn_mfma_ok(dt, d, q, k, v, d_o) && (((uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv) & 7) == 0)
        return vaw_attn_bwd_mfma(d, q, k, v, o, d_o, lse, delta, dq, dk, dv, s);
    AttnDev a = to_dev(d);
    dim3 grid(ceil_div(a.T, 4), a.B * a.H);
    const size_t lds_q = 4 * (size_t)(2 * a.hd + a.T) * sizeof(float);
    const size_t lds_kv = 4 * (size_t)(2 * a.hd + 2 * a.T) * sizeof(float);
    if (dt == VAW_F32) {
        attn_bwd_q_rowwise<float><<<grid, 256, lds_q, s>>>(a, (const float*)q, (const float*)k, (const float*)v, (const float*)o, (const float*)d_o, lse, delta, (float*)dq);
        attn_bwd_kv_rowwise<float><<<grid, 256, lds_kv, s>>>(a, (const float*)q, (const float*)k, (const float*)v, (const float*)d_o, lse, delta, (float*)dk, (float*)dv);
    } else {
        attn_bwd_q_rowwise<bf16_t><<<grid, 256, lds_q, s>>>(a, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (const bf16_t*)o, (const bf16_t*)d_o, lse, delta, (bf16_t*)dq);
        attn_bwd_kv_rowwise<bf16_t><<<grid, 256, lds_kv, s>>>(a, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (const bf16_t*)d_o, lse, delta, (bf16_t*)dk, (bf16_t*)dv);
    }
    VAW_CHECK_LAUNCH("attn_bwd");
    return VAW_OK;
}

// vaw_attn_bwd that also leaves the column sums of dq | dk | dv behind as partial rows (packed-qkv column order [3][H][hd], the
// bias layout of timm Attention's qkv Linear): colsum_partial [*rows_out][3 H hd] f32, at most B * T / 64 rows.  Only the bf16
// MFMA kernels offer it: VAW_ERR_UNSUPPORTED (nothing launched) otherwise -- call vaw_attn_bwd and vaw_colsum then.
extern "C" int vaw_attn_bwd_colsum(vaw_dtype dt, const vaw_attn_desc* d, const void* q, const void* k, const void* v, const void* o,
                                   const void* d_o, const float* lse, float* delta, void* dq, void* dk, void* dv,
                                   float* colsum_partial, int64_t* rows_out, vaw_stream stream) {
    int rc = check_desc(d, "attn_bwd_colsum");
    if (rc) return rc;
    VAW_CHECK_ARG(colsum_partial && rows_out, "attn_bwd_colsum: colsum_partial and rows_out are required");
    VAW_CHECK_ARG(*rows_out >= (int64_t)d->B * (d->T / 64 > 0 ? d->T / 64 : 1), "attn_bwd_colsum: *rows_out states a capacity of %ld rows, up to %ld are written",
                  (long)*rows_out, (long)((int64_t)d->B * (d->T / 64 > 0 ? d->T / 64 : 1)));
    if (!g_force_rowwise && vaw_attn_mfma_ok(dt, d, q, k, v, d_o) && (((uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv) & 7) == 0 && d->hd % 4 == 0)
        return vaw_attn_bwd_mfma(d, q, k, v, o, d_o, lse, delta, dq, dk, dv, (hipStream_t)stream, colsum_partial, rows_out);
    vaw_set_error("attn_bwd_colsum: only the bf16 MFMA attention kernels carry column sums");
    return VAW_ERR_UNSUPPORTED;
}
