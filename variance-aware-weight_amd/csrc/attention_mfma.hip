// bf16 MFMA attention for the token-major DiT layout, head dim 64, T in {64,128,192,256}.
//
// Everything is computed TRANSPOSED so that no probability tile ever crosses lanes or LDS:
//   S^T[key][query] = K . Q^T  lands in the 16x16 accumulator layout with the QUERY on the lane (col = l&15) and
//   the keys in registers (row = 4(l>>4)+r); softmax statistics are then an in-lane loop plus two shuffles
//   (xor 16, 32), and bf16(P^T) is ALREADY the B operand of the next product O^T = V^T . P^T
//   (cdna_hip_programming.md §3 "An accumulator tile as the next MFMA's operand": the k order of such an operand
//   is permuted -- k slot (g, jj) holds accumulator row 4g+jj of tile 2s (jj<4) or tile 2s+1 (jj>=4) -- so the A
//   operand V^T is fetched with the same permutation by ds_read_b64_tr_b16).
// Backward recomputes P from the saved row log-sum-exp in BOTH orientations (S^T for dQ, S for dK/dV): two
// extra 64x64x64 products per tile buy a kernel with no transposes, no atomics and no T x T tensor in memory.
// delta_i = sum_j P_ij dP_ij is taken from the tiles themselves (equals rowsum(dO*O)).
//
// LDS: every operand tile is a [rows][64] bf16 image with 128-byte rows, filled by LDS-DMA
// (global_load_lds_dwordx4, 8 rows per wave-instruction) with the 16-byte chunk XOR-swizzled on the SOURCE
// side: chunk' = chunk ^ ((row>>1)&7).  Row reads (ds_read_b128) are conflict-free, transposed reads 2-way.
#include "common.h"

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

struct AttnMfmaArgs {
    int B, H, T;
    int64_t q_sb, q_sh, q_st;   // element strides of q/k/v (and dq/dk/dv); channel stride is 1
    int64_t o_sb, o_sh, o_st;   // element strides of o / d_o
    float scale;
};

__device__ __forceinline__ int img_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// rows [r_begin, r_begin+nrows) of a token-major operand -> LDS image rows [0, nrows); all 4 waves cooperate
__device__ __forceinline__ void stage_rows(const bf16_t* __restrict__ g, int64_t stride_t, int nrows, char* img, int wid,
                                           int lane) {
    for (int inst = wid; inst < nrows / 8; inst += 4) {
        const int row = inst * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(g + (int64_t)row * stride_t + chunk * 8),
                                         (lds_ptr_t)(img + inst * 1024), 16, 0, 0);
    }
}

// 16 rows x 32 k (k = channel), rows r0.., k-step s: the natural A (or B) fragment of a [rows][64] image
__device__ __forceinline__ bf16x8 frag_rows(const char* img, int r0, int s, int lane) {
    return *reinterpret_cast<const bf16x8*>(img + img_off(r0 + (lane & 15), 4 * s + (lane >> 4)));
}
// Transposed fragment: operand row = image column d0 + (l&15), k = image rows in the accumulator-derived order
// {kbase + 4g + 0..3, kbase + 16 + 4g + 0..3}, g = l>>4.
__device__ __forceinline__ bf16x8 frag_cols_perm(const char* img, int d0, int kbase, int lane) {
    const int li = lane & 15, q = li >> 2, p = li & 3, g = lane >> 4;
    const int ch = (d0 >> 3) + (p >> 1);
    const int r_lo = kbase + 4 * g + q, r_hi = r_lo + 16;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(img + img_off(r_lo, ch) + 8 * (p & 1)));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(img + img_off(r_hi, ch) + 8 * (p & 1)));
    bf16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return r;
}
__device__ __forceinline__ bf16x8 pack_acc(f32x4 a, f32x4 b) {
    bf16x8 r = {(bf16_t)a[0], (bf16_t)a[1], (bf16_t)a[2], (bf16_t)a[3], (bf16_t)b[0], (bf16_t)b[1], (bf16_t)b[2], (bf16_t)b[3]};
    return r;
}
__device__ __forceinline__ float group_sum(float v) {   // over the 4 lanes l, l^16, l^32, l^48
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
__device__ __forceinline__ float group_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    v = fmaxf(v, __shfl_xor(v, 32, 64));
    return v;
}
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)

// ------------------------------------------------------------------------------------------------
// forward: grid (T/64, B*H), 4 waves x 16 queries.  NT = T/16 key tiles.
// ------------------------------------------------------------------------------------------------
template <int NT>
__global__ void __launch_bounds__(256)
attn_fwd_mfma(AttnMfmaArgs a, const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
              bf16_t* __restrict__ o, float* __restrict__ lse) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int T = NT * 16;
    char* qimg = smem;                  // [64][64]
    char* kimg = smem + 64 * 128;       // [T][64]
    char* vimg = kimg + T * 128;        // [T][64]
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int bh = blockIdx.y, b = bh / a.H, h = bh % a.H;
    const int qb = blockIdx.x * 64;
    const int64_t base = b * a.q_sb + h * a.q_sh;
    stage_rows(q + base + (int64_t)qb * a.q_st, a.q_st, 64, qimg, wid, lane);
    stage_rows(k + base, a.q_st, T, kimg, wid, lane);
    stage_rows(v + base, a.q_st, T, vimg, wid, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    f32x4 st[NT];
    const bf16x8 qf0 = frag_rows(qimg, 16 * wid, 0, lane), qf1 = frag_rows(qimg, 16 * wid, 1, lane);
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) {
        f32x4 c = {0, 0, 0, 0};
        c = MFMA(frag_rows(kimg, 16 * jt, 0, lane), qf0, c);
        c = MFMA(frag_rows(kimg, 16 * jt, 1, lane), qf1, c);
        st[jt] = c;
    }
    float mx = -INFINITY;
#pragma unroll
    for (int jt = 0; jt < NT; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st[jt][r]);
    mx = group_max(mx) * a.scale;
    float l = 0.f;
#pragma unroll
    for (int jt = 0; jt < NT; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            st[jt][r] = __expf(st[jt][r] * a.scale - mx);
            l += st[jt][r];
        }
    l = group_sum(l);
    f32x4 ot[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) ot[dt] = f32x4{0, 0, 0, 0};
#pragma unroll
    for (int s2 = 0; s2 < NT / 2; ++s2) {
        const bf16x8 pf = pack_acc(st[2 * s2], st[2 * s2 + 1]);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) ot[dt] = MFMA(frag_cols_perm(vimg, 16 * dt, 32 * s2, lane), pf, ot[dt]);
    }
    const float inv = 1.f / l;
    const int qi = qb + 16 * wid + (lane & 15);
    bf16_t* orow = o + b * a.o_sb + h * a.o_sh + (int64_t)qi * a.o_st + 4 * (lane >> 4);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) store4(orow + 16 * dt, ot[dt] * inv);
    if ((lane >> 4) == 0) lse[(int64_t)bh * a.T + qi] = mx + __logf(l);
}

// ------------------------------------------------------------------------------------------------
// backward: grid (B*H), 4 waves; phase A = query-major (dQ, delta), phase B = key-major (dK, dV).
// ------------------------------------------------------------------------------------------------
template <int NT>
__global__ void __launch_bounds__(256)
attn_bwd_mfma(AttnMfmaArgs a, const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
              const bf16_t* __restrict__ d_o, const float* __restrict__ lse, float* __restrict__ delta_out,
              bf16_t* __restrict__ dq, bf16_t* __restrict__ dk, bf16_t* __restrict__ dv) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int T = NT * 16;
    char* qimg = smem;
    char* kimg = qimg + T * 128;
    char* vimg = kimg + T * 128;
    char* gimg = vimg + T * 128;                                    // dO
    float* lse_s = reinterpret_cast<float*>(gimg + T * 128);        // [T]
    float* del_s = lse_s + T;                                       // [T]
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int bh = blockIdx.x, b = bh / a.H, h = bh % a.H;
    const int64_t base = b * a.q_sb + h * a.q_sh, obase = b * a.o_sb + h * a.o_sh;
    stage_rows(q + base, a.q_st, T, qimg, wid, lane);
    stage_rows(k + base, a.q_st, T, kimg, wid, lane);
    stage_rows(v + base, a.q_st, T, vimg, wid, lane);
    stage_rows(d_o + obase, a.o_st, T, gimg, wid, lane);
    for (int i = threadIdx.x; i < T; i += 256) lse_s[i] = lse[(int64_t)bh * a.T + i];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int g = lane >> 4, li = lane & 15;

    // ---- phase A: this wave's queries are the lane dimension ---------------------------------
    for (int qt = wid; qt < NT; qt += 4) {
        const int q0 = 16 * qt;
        const bf16x8 qf0 = frag_rows(qimg, q0, 0, lane), qf1 = frag_rows(qimg, q0, 1, lane);
        const bf16x8 gf0 = frag_rows(gimg, q0, 0, lane), gf1 = frag_rows(gimg, q0, 1, lane);
        f32x4 p[NT], dp[NT];
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) {
            f32x4 c = {0, 0, 0, 0}, d = {0, 0, 0, 0};
            c = MFMA(frag_rows(kimg, 16 * jt, 0, lane), qf0, c);
            c = MFMA(frag_rows(kimg, 16 * jt, 1, lane), qf1, c);
            d = MFMA(frag_rows(vimg, 16 * jt, 0, lane), gf0, d);
            d = MFMA(frag_rows(vimg, 16 * jt, 1, lane), gf1, d);
            p[jt] = c;
            dp[jt] = d;
        }
        const float li_lse = lse_s[q0 + li];
        float dl = 0.f;
#pragma unroll
        for (int jt = 0; jt < NT; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                p[jt][r] = __expf(p[jt][r] * a.scale - li_lse);
                dl += p[jt][r] * dp[jt][r];
            }
        dl = group_sum(dl);
        if (g == 0) {
            del_s[q0 + li] = dl;
            delta_out[(int64_t)bh * a.T + q0 + li] = dl;
        }
#pragma unroll
        for (int jt = 0; jt < NT; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) p[jt][r] = a.scale * p[jt][r] * (dp[jt][r] - dl);   // dS^T (scaled)
        f32x4 acc[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) acc[dt] = f32x4{0, 0, 0, 0};
#pragma unroll
        for (int s2 = 0; s2 < NT / 2; ++s2) {
            const bf16x8 sf = pack_acc(p[2 * s2], p[2 * s2 + 1]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) acc[dt] = MFMA(frag_cols_perm(kimg, 16 * dt, 32 * s2, lane), sf, acc[dt]);
        }
        bf16_t* row = dq + base + (int64_t)(q0 + li) * a.q_st + 4 * g;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) store4(row + 16 * dt, acc[dt]);
    }
    __syncthreads();   // delta of every query is in LDS

    // ---- phase B: this wave's keys are the lane dimension ---------------------------------
    for (int kt = wid; kt < NT; kt += 4) {
        const int j0 = 16 * kt;
        const bf16x8 kf0 = frag_rows(kimg, j0, 0, lane), kf1 = frag_rows(kimg, j0, 1, lane);
        const bf16x8 vf0 = frag_rows(vimg, j0, 0, lane), vf1 = frag_rows(vimg, j0, 1, lane);
        f32x4 p[NT], ds[NT];
#pragma unroll
        for (int it = 0; it < NT; ++it) {
            f32x4 c = {0, 0, 0, 0}, d = {0, 0, 0, 0};
            c = MFMA(frag_rows(qimg, 16 * it, 0, lane), kf0, c);
            c = MFMA(frag_rows(qimg, 16 * it, 1, lane), kf1, c);
            d = MFMA(frag_rows(gimg, 16 * it, 0, lane), vf0, d);
            d = MFMA(frag_rows(gimg, 16 * it, 1, lane), vf1, d);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * it + 4 * g + r;
                const float pr = __expf(c[r] * a.scale - lse_s[i]);
                c[r] = pr;
                d[r] = a.scale * pr * (d[r] - del_s[i]);
            }
            p[it] = c;
            ds[it] = d;
        }
        f32x4 av[4], ak[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) av[dt] = ak[dt] = f32x4{0, 0, 0, 0};
#pragma unroll
        for (int s2 = 0; s2 < NT / 2; ++s2) {
            const bf16x8 pf = pack_acc(p[2 * s2], p[2 * s2 + 1]);
            const bf16x8 sf = pack_acc(ds[2 * s2], ds[2 * s2 + 1]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                av[dt] = MFMA(frag_cols_perm(gimg, 16 * dt, 32 * s2, lane), pf, av[dt]);
                ak[dt] = MFMA(frag_cols_perm(qimg, 16 * dt, 32 * s2, lane), sf, ak[dt]);
            }
        }
        const int64_t off = base + (int64_t)(j0 + li) * a.q_st + 4 * g;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            store4(dv + off + 16 * dt, av[dt]);
            store4(dk + off + 16 * dt, ak[dt]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host dispatch (called from attention.hip)
// ------------------------------------------------------------------------------------------------
static bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

bool vaw_attn_mfma_ok(vaw_dtype dt, const vaw_attn_desc* d, const void* q, const void* k, const void* v, const void* o) {
    return dt == VAW_BF16 && d->hd == 64 && d->T % 64 == 0 && d->T <= 256 && d->q_sd == 1 && d->o_sd == 1 &&
           d->q_st % 8 == 0 && d->q_sh % 8 == 0 && d->q_sb % 8 == 0 && d->o_st % 8 == 0 && d->o_sh % 8 == 0 &&
           d->o_sb % 8 == 0 && aligned16(q) && aligned16(k) && aligned16(v) && aligned16(o) && (int64_t)d->B * d->H < 65536;
}

static AttnMfmaArgs mk_args(const vaw_attn_desc* d) {
    AttnMfmaArgs a{d->B, d->H, d->T, d->q_sb, d->q_sh, d->q_st, d->o_sb, d->o_sh, d->o_st, d->scale};
    return a;
}

#define DISPATCH_NT(T, ...)                               \
    switch ((T) / 16) {                                   \
        case 4: { constexpr int NT = 4; __VA_ARGS__ } break;   \
        case 8: { constexpr int NT = 8; __VA_ARGS__ } break;   \
        case 12: { constexpr int NT = 12; __VA_ARGS__ } break; \
        default: { constexpr int NT = 16; __VA_ARGS__ } break; \
    }

int vaw_attn_fwd_mfma(const vaw_attn_desc* d, const void* q, const void* k, const void* v, void* o, float* lse,
                      hipStream_t s) {
    AttnMfmaArgs a = mk_args(d);
    dim3 grid(d->T / 64, d->B * d->H);
    const size_t lds = (size_t)(64 + 2 * d->T) * 128;
    DISPATCH_NT(d->T,
        (void)hipFuncSetAttribute((const void*)attn_fwd_mfma<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attn_fwd_mfma<NT><<<grid, 256, lds, s>>>(a, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)o, lse);
    )
    VAW_CHECK_LAUNCH("attn_fwd_mfma");
    return VAW_OK;
}

int vaw_attn_bwd_mfma(const vaw_attn_desc* d, const void* q, const void* k, const void* v, const void* d_o,
                      const float* lse, float* delta, void* dq, void* dk, void* dv, hipStream_t s) {
    AttnMfmaArgs a = mk_args(d);
    const size_t lds = (size_t)4 * d->T * 128 + 2 * d->T * sizeof(float);
    DISPATCH_NT(d->T,
        (void)hipFuncSetAttribute((const void*)attn_bwd_mfma<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attn_bwd_mfma<NT><<<d->B * d->H, 256, lds, s>>>(a, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v,
                                                          (const bf16_t*)d_o, lse, delta, (bf16_t*)dq, (bf16_t*)dk, (bf16_t*)dv);
    )
    VAW_CHECK_LAUNCH("attn_bwd_mfma");
    return VAW_OK;
}
