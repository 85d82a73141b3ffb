// fp8 (OCP e4m3fn) operands for the Linear GEMMs of DiT -- BASELINE.json config 5 ("DiT-XL/2, fp8 MFMA GEMMs + bf16 accum";
// reference recipe run.sh:20-26, model models/dit.py:373).
//
//   vaw_fp8_quantize   per-tensor scaling: q = e4m3(x * 448 / amax(|x|)), written row-major AND (optionally) transposed, with the
//                      dequantisation scale amax / 448 left in device memory (nothing syncs with the host).  The transposed copy is
//                      what lets every GEMM of the step run k-major x k-major: dgrad reads W^T, wgrad reads dy^T and x^T.
//   vaw_gemm_fp8       C = epilogue(alpha * scale_a * scale_b * A[M,K] . B[N,K]^T) on the persistent kernel of gemm_p8_kernel.h with
//                      v_mfma_scale_f32_16x16x128_f8f6f4 (unit block scales): twice the bf16 MFMA rate, half the staged bytes.  Same
//                      epilogues (bias / GELU / GELU' / gated residual / column sums), f32 accumulation, bf16 or f32 output.
//   (vaw_wgrad_grouped with dt = VAW_FP8 runs the deferred weight gradients on the transposed copies.)
#include "gemm_p8_kernel.h"

struct P8Plan {
    bool use;
    int ntw, split, grid;
};
P8Plan vaw_p8_plan(int64_t M, int64_t N, int64_t K, bool plain_f32, bool want_colsum, int64_t ws_floats, int force);
extern "C" int vaw_reduce_rows(const float* partial, int64_t R, int64_t N, float* out, float beta, vaw_stream stream);

// a_e5m2: the dy operand is e5m2 (vaw_wgrad_grouped with dt = VAW_BF8); x is e4m3 either way
void vaw_p8_group_fp8(int ntw, bool a_e5m2, int nk, int grid, const EpiDev& e, const P8Prob* probs_dev, const P8Group& grp, hipStream_t s) {
    if (a_e5m2) {
        if (ntw == 4) p8_launch_group<4, 2>(nk, grid, e, probs_dev, grp, s);
        else p8_launch_group<3, 2>(nk, grid, e, probs_dev, grp, s);
    } else {
        if (ntw == 4) p8_launch_group<4, 1>(nk, grid, e, probs_dev, grp, s);
        else p8_launch_group<3, 1>(nk, grid, e, probs_dev, grp, s);
    }
}

// ---- quantisation ---------------------------------------------------------------------------------------------------
// |x| maxima: max is exact and order-independent, so per-workgroup partials folded by a second launch are deterministic.
template <typename T>
__global__ void fp8_amax_partial_kernel(const T* __restrict__ x, int64_t R, int64_t C, int64_t ld, float* __restrict__ part) {
    float m = 0.f;
    const int64_t n4 = R * (C / 4), c4n = C / 4;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const f32x4 v = load4(x + (i / c4n) * ld + (i % c4n) * 4);
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    }
    __shared__ float sh[4];
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
}
__global__ void fp8_amax_final_kernel(const float* __restrict__ part, int n, float* __restrict__ scale, float fmt_max) {
    float m = 0.f;
    for (int i = threadIdx.x; i < n; i += 64) m = fmaxf(m, part[i]);
    m = wave_max(m);
    if (threadIdx.x == 0) scale[0] = m > 0.f ? m / fmt_max : 1.f;    // dequantisation scale; fmt_max = largest finite value of the format
}

template <bool E5M2>
__device__ __forceinline__ unsigned fp8_pack4(f32x4 v) {
    unsigned r = 0;
    if (E5M2) {
        r = __builtin_amdgcn_cvt_pk_bf8_f32(v[0], v[1], r, false);   // v_cvt_pk_bf8_f32: OCP e5m2 on gfx950, round to nearest even
        r = __builtin_amdgcn_cvt_pk_bf8_f32(v[2], v[3], r, true);
    } else {
        r = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], r, false);   // v_cvt_pk_fp8_f32: OCP e4m3fn
        r = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], r, true);
    }
    return r;
}

// One 64 x 64 tile per workgroup: q rows straight out (coalesced 4-byte stores), the transposed copy through LDS.
// amax_acc != NULL (delayed scaling): `scale` is the one fixed before this step; values beyond its range saturate, and the
// tensor's max |x| is folded into *amax_acc with an integer atomic max on the float's bits (non-negative floats order like
// unsigned integers; max is order-independent, so the result does not depend on scheduling).
template <typename T, bool E5M2>
__device__ __forceinline__ void fp8_quantize_tile(const T* __restrict__ x, int64_t R, int64_t C, int64_t ld, unsigned char* __restrict__ q, int64_t ldq,
                                                  unsigned char* __restrict__ qt, int64_t ldt, const float* __restrict__ scale,
                                                  float* __restrict__ amax_acc, int bx, int by) {
    __shared__ unsigned char tile[64][68];
    __shared__ float sh[4];
    const float inv = 1.f / scale[0];
    const float fmax_ = E5M2 ? 57344.f : 448.f;
    const int64_t r0 = (int64_t)by * 64, c0 = (int64_t)bx * 64;
    const int tr = threadIdx.x >> 4, tc = (threadIdx.x & 15) * 4;              // 16 rows x 16 groups of 4 columns per pass
    float am = 0.f;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int r = pass * 16 + tr;
        unsigned w = 0;
        if (r0 + r < R && c0 + tc < C) {                                       // C % 4 == 0: a group of 4 is in or out as a whole
            f32x4 v = load4(x + (r0 + r) * ld + c0 + tc);
            if (amax_acc) {
                am = fmaxf(fmaxf(am, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
                v = v * inv;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = __builtin_amdgcn_fmed3f(v[j], -fmax_, fmax_);
            } else {
                v = v * inv;
            }
            w = fp8_pack4<E5M2>(v);
            *reinterpret_cast<unsigned*>(q + (r0 + r) * ldq + c0 + tc) = w;
        }
        if (qt) *reinterpret_cast<unsigned*>(&tile[r][tc]) = w;
    }
    if (amax_acc) {
        am = wave_max(am);
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = am;
    }
    __syncthreads();
    if (amax_acc && threadIdx.x == 0) {
        const float m = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
        // look first: the running max only grows, so after the first few workgroups almost nobody needs the atomic (tens of
        // thousands of same-address atomics per tensor cost more than the pass they replace); a stale read only costs a spare atomic
        unsigned* acc = reinterpret_cast<unsigned*>(amax_acc);
        const unsigned mb = __float_as_uint(m);
        if (m > 0.f && !(m != m) && mb > __hip_atomic_load(acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(acc, mb);
    }
    if (!qt) return;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {                                     // out row = input column
        const int c = pass * 16 + tr;
        if (c0 + c < C && r0 + tc < R) {
            const unsigned w = (unsigned)tile[tc][c] | ((unsigned)tile[tc + 1][c] << 8) | ((unsigned)tile[tc + 2][c] << 16) |
                               ((unsigned)tile[tc + 3][c] << 24);
            if (r0 + tc + 3 < R) *reinterpret_cast<unsigned*>(qt + (c0 + c) * ldt + r0 + tc) = w;
            else
                for (int j = 0; j < 4 && r0 + tc + j < R; ++j) qt[(c0 + c) * ldt + r0 + tc + j] = tile[tc + j][c];
        }
    }
}

template <typename T, bool E5M2>
__global__ void fp8_quantize_kernel(const T* __restrict__ x, int64_t R, int64_t C, int64_t ld, unsigned char* __restrict__ q, int64_t ldq,
                                    unsigned char* __restrict__ qt, int64_t ldt, const float* __restrict__ scale, float* __restrict__ amax_acc) {
    fp8_quantize_tile<T, E5M2>(x, R, C, ld, q, ldq, qt, ldt, scale, amax_acc, (int)blockIdx.x, (int)blockIdx.y);
}

// MANY f32 tensors in one launch (delayed scaling, e4m3): the weights of every Linear layer of the blocks are re-quantised from
// their f32 masters once per optimizer step -- 112 tensors for DiT-XL, each an 18-microsecond launch of its own before (2 ms of
// the fp8 step; the tensors are 1-5 M elements: latency-, not bandwidth-bound).  Job j owns workgroups [block0_j, block0_{j+1}).
struct QuantJobDev {
    const float* src;
    unsigned char *q, *qt;
    float* state;
    int64_t R, C, ld, ldq, ldt;
    int block0, tiles_x;
};
__global__ void fp8_quantize_batched_kernel(const QuantJobDev* __restrict__ jobs, int n_jobs) {
    int lo = 0, hi = n_jobs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].block0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const QuantJobDev jb = jobs[lo];
    const int t = (int)blockIdx.x - jb.block0;
    fp8_quantize_tile<float, false>(jb.src, jb.R, jb.C, jb.ld, jb.q, jb.ldq, jb.qt, jb.ldt, jb.state, jb.state + 1, t % jb.tiles_x, t / jb.tiles_x);
}

// The training step's shapes (bf16 source, R % 64 == 0, C % 128 == 0, both copies wanted, 16-byte aligned rows): 64 x 128
// tile per workgroup, 16-byte loads, 8-byte q stores (128-byte row segments), and the transposed copy by WORDS: the tile goes
// to LDS as 4-byte groups, each lane reads a 4 x 4 byte block (four words of four consecutive rows), transposes it in
// registers with v_perm_b32 and stores four 4-byte groups of qt (16 lanes = one 64-byte row segment).  A quarter of the LDS
// operations of the byte-gathering kernel above.
template <bool E5M2>
__global__ void __launch_bounds__(256)
fp8_quantize_bf16_tile_kernel(const bf16_t* __restrict__ x, int64_t ld, unsigned char* __restrict__ q, int64_t ldq,
                              unsigned char* __restrict__ qt, int64_t ldt, const float* __restrict__ scale, float* __restrict__ amax_acc) {
    __shared__ unsigned tile[64][33];                      // [row][word of 4 columns], +1 word of padding
    __shared__ float sh[4];
    const float inv = 1.f / scale[0];
    const float fmax_ = E5M2 ? 57344.f : 448.f;
    const int64_t r0 = (int64_t)blockIdx.y * 64, c0 = (int64_t)blockIdx.x * 128;
    const int tr = threadIdx.x >> 4, c8 = threadIdx.x & 15;
    float am = 0.f;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int r = pass * 16 + tr;
        const bf16x8 h = *reinterpret_cast<const bf16x8*>(x + (r0 + r) * ld + c0 + 8 * c8);
        f32x4 v0 = {(float)h[0], (float)h[1], (float)h[2], (float)h[3]}, v1 = {(float)h[4], (float)h[5], (float)h[6], (float)h[7]};
        if (amax_acc) {
#pragma unroll
            for (int j = 0; j < 4; ++j) am = fmaxf(am, fmaxf(fabsf(v0[j]), fabsf(v1[j])));
            v0 = v0 * inv; v1 = v1 * inv;
#pragma unroll
            for (int j = 0; j < 4; ++j) { v0[j] = __builtin_amdgcn_fmed3f(v0[j], -fmax_, fmax_); v1[j] = __builtin_amdgcn_fmed3f(v1[j], -fmax_, fmax_); }
        } else {
            v0 = v0 * inv; v1 = v1 * inv;
        }
        const unsigned w0 = fp8_pack4<E5M2>(v0), w1 = fp8_pack4<E5M2>(v1);
        *reinterpret_cast<uint2*>(q + (r0 + r) * ldq + c0 + 8 * c8) = uint2{w0, w1};
        tile[r][2 * c8] = w0;
        tile[r][2 * c8 + 1] = w1;
    }
    if (amax_acc) {
        am = wave_max(am);
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = am;
    }
    __syncthreads();
    if (amax_acc && threadIdx.x == 0) {
        const float m = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
        unsigned* acc = reinterpret_cast<unsigned*>(amax_acc);
        const unsigned mb = __float_as_uint(m);
        if (m > 0.f && !(m != m) && mb > __hip_atomic_load(acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(acc, mb);
    }
    const int rg = threadIdx.x & 15, cq0 = threadIdx.x >> 4;      // rows 4 rg .. 4 rg + 3; column quads cq0, cq0 + 16
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int cq = cq0 + 16 * pass;
        const unsigned a = tile[4 * rg][cq], b = tile[4 * rg + 1][cq], c = tile[4 * rg + 2][cq], d = tile[4 * rg + 3][cq];
        // v_perm_b32(hi, lo, sel): result byte i = byte sel[i] of the 8-byte value {hi, lo} (lo = bytes 0-3)
        const unsigned ab_lo = __builtin_amdgcn_perm(b, a, 0x05010400), ab_hi = __builtin_amdgcn_perm(b, a, 0x07030602);   // a0 b0 a1 b1 | a2 b2 a3 b3
        const unsigned cd_lo = __builtin_amdgcn_perm(d, c, 0x05010400), cd_hi = __builtin_amdgcn_perm(d, c, 0x07030602);
        const unsigned o0 = __builtin_amdgcn_perm(cd_lo, ab_lo, 0x05040100), o1 = __builtin_amdgcn_perm(cd_lo, ab_lo, 0x07060302);
        const unsigned o2 = __builtin_amdgcn_perm(cd_hi, ab_hi, 0x05040100), o3 = __builtin_amdgcn_perm(cd_hi, ab_hi, 0x07060302);
        unsigned char* dst = qt + (c0 + 4 * cq) * ldt + r0 + 4 * rg;
        *reinterpret_cast<unsigned*>(dst) = o0;
        *reinterpret_cast<unsigned*>(dst + ldt) = o1;
        *reinterpret_cast<unsigned*>(dst + 2 * ldt) = o2;
        *reinterpret_cast<unsigned*>(dst + 3 * ldt) = o3;
    }
}

// qt[c][r] = q[r][c] for fp8 bytes (R % 64 == 0, C % 128 == 0): the transposed copy of a tensor that a GEMM epilogue already
// wrote as fp8 (vaw_gemm_fp8 with c_fp8_state).  Same word-wise LDS transposition as fp8_quantize_bf16_tile_kernel.
__global__ void __launch_bounds__(256)
fp8_transpose_kernel(const unsigned char* __restrict__ q, int64_t ldq, unsigned char* __restrict__ qt, int64_t ldt) {
    __shared__ unsigned tile[64][33];
    const int64_t r0 = (int64_t)blockIdx.y * 64, c0 = (int64_t)blockIdx.x * 128;
    const int tr = threadIdx.x >> 4, c8 = threadIdx.x & 15;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int r = pass * 16 + tr;
        const uint2 w = *reinterpret_cast<const uint2*>(q + (r0 + r) * ldq + c0 + 8 * c8);
        tile[r][2 * c8] = w.x;
        tile[r][2 * c8 + 1] = w.y;
    }
    __syncthreads();
    const int rg = threadIdx.x & 15, cq0 = threadIdx.x >> 4;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int cq = cq0 + 16 * pass;
        const unsigned a = tile[4 * rg][cq], b = tile[4 * rg + 1][cq], c = tile[4 * rg + 2][cq], d = tile[4 * rg + 3][cq];
        const unsigned ab_lo = __builtin_amdgcn_perm(b, a, 0x05010400), ab_hi = __builtin_amdgcn_perm(b, a, 0x07030602);
        const unsigned cd_lo = __builtin_amdgcn_perm(d, c, 0x05010400), cd_hi = __builtin_amdgcn_perm(d, c, 0x07030602);
        unsigned char* dst = qt + (c0 + 4 * cq) * ldt + r0 + 4 * rg;
        *reinterpret_cast<unsigned*>(dst) = __builtin_amdgcn_perm(cd_lo, ab_lo, 0x05040100);
        *reinterpret_cast<unsigned*>(dst + ldt) = __builtin_amdgcn_perm(cd_lo, ab_lo, 0x07060302);
        *reinterpret_cast<unsigned*>(dst + 2 * ldt) = __builtin_amdgcn_perm(cd_hi, ab_hi, 0x05040100);
        *reinterpret_cast<unsigned*>(dst + 3 * ldt) = __builtin_amdgcn_perm(cd_hi, ab_hi, 0x07060302);
    }
}

// The same for R % 128 == 0 and 16-byte aligned rows on both sides: 128 x 128-byte tiles, 16 bytes per lane in AND out, and the
// output leaves as WHOLE 128-byte rows.  The 64 x 128 kernels above write 4 bytes per lane in 64-byte runs and reach 2.5 TB/s; a first
// 128 x 128 version stored its transposed registers directly (16 bytes per lane, but 32-byte runs of 32 different rows per
// wave-instruction: 29.5 us for the 37.7 MB tensors of DiT-XL/2, still 2.5 TB/s).  Now the transposed 16-byte pieces go back into
// the LDS tile (XOR-swizzled chunks: conflict-free for the writes and the row reads) and eight lanes store one output row.
//   tile in : [source row][33 words]            (+1 word of padding)
//   tile out: [output row o][8 chunks of 16 B], chunk k (= source rows 16 k .. 16 k + 15) at position k ^ ((o >> 2) & 7)
// Thread (cq = 4-column group, r16 = 16-row group) gathers the 16 x 4 byte block rows r16 .. r16 + 15, columns 4 cq .. 4 cq + 3
// from LDS (16 words) and transposes it with v_perm_b32 (four 4 x 4 byte transposes).
__device__ __forceinline__ void fp8_tile128_transpose_store(unsigned (*tile)[33], unsigned char* __restrict__ qt, int64_t ldt,
                                                            int64_t r0, int64_t c0) {
    const int cq = threadIdx.x & 31, k16 = threadIdx.x >> 5, r16 = k16 * 16;
    unsigned o[4][4];                                     // o[column j of the group][word = rows 4 i .. 4 i + 3]
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned a = tile[r16 + 4 * i][cq], b = tile[r16 + 4 * i + 1][cq], c = tile[r16 + 4 * i + 2][cq], d = tile[r16 + 4 * i + 3][cq];
        const unsigned ab_lo = __builtin_amdgcn_perm(b, a, 0x05010400), ab_hi = __builtin_amdgcn_perm(b, a, 0x07030602);
        const unsigned cd_lo = __builtin_amdgcn_perm(d, c, 0x05010400), cd_hi = __builtin_amdgcn_perm(d, c, 0x07030602);
        o[0][i] = __builtin_amdgcn_perm(cd_lo, ab_lo, 0x05040100);
        o[1][i] = __builtin_amdgcn_perm(cd_lo, ab_lo, 0x07060302);
        o[2][i] = __builtin_amdgcn_perm(cd_hi, ab_hi, 0x05040100);
        o[3][i] = __builtin_amdgcn_perm(cd_hi, ab_hi, 0x07060302);
    }
    __syncthreads();                                      // everybody has read the input tile: reuse its memory for the output
    uint4* out = reinterpret_cast<uint4*>(&tile[0][0]);   // [128 rows][8 chunks]
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int orow = 4 * cq + j;
        out[orow * 8 + (k16 ^ (cq & 7))] = uint4{o[j][0], o[j][1], o[j][2], o[j][3]};
    }
    __syncthreads();
    const int ch = threadIdx.x & 7, rr = threadIdx.x >> 3;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int orow = pass * 32 + rr;
        *reinterpret_cast<uint4*>(qt + (c0 + orow) * ldt + r0 + 16 * ch) = out[orow * 8 + (ch ^ ((orow >> 2) & 7))];
    }
}

__global__ void __launch_bounds__(256)
fp8_transpose128_kernel(const unsigned char* __restrict__ q, int64_t ldq, unsigned char* __restrict__ qt, int64_t ldt) {
    __shared__ __attribute__((aligned(16))) unsigned tile[128][33];                    // [source row][word column], +1 word of padding
    const int64_t r0 = (int64_t)blockIdx.y * 128, c0 = (int64_t)blockIdx.x * 128;
    {
        const int ch = threadIdx.x & 7, rr = threadIdx.x >> 3;            // 16-byte chunk of the row, row within the pass
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int r = pass * 32 + rr;
            const uint4 w = *reinterpret_cast<const uint4*>(q + (r0 + r) * ldq + c0 + 16 * ch);
            tile[r][4 * ch] = w.x; tile[r][4 * ch + 1] = w.y; tile[r][4 * ch + 2] = w.z; tile[r][4 * ch + 3] = w.w;
        }
    }
    __syncthreads();
    fp8_tile128_transpose_store(tile, qt, ldt, r0, c0);
}

// bf16 -> fp8 bytes q AND their transposed copy qt on the same 128 x 128 tile (R % 128 == 0, C % 128 == 0, 16-byte aligned rows):
// 32 bytes in, 16 bytes of q out per lane and row piece; qt as in fp8_transpose128_kernel.  Same values as
// fp8_quantize_bf16_tile_kernel (same scale, saturation, rounding and running-max update).
template <bool E5M2>
__global__ void __launch_bounds__(256)
fp8_quantize_bf16_tile128_kernel(const bf16_t* __restrict__ x, int64_t ld, unsigned char* __restrict__ q, int64_t ldq,
                                 unsigned char* __restrict__ qt, int64_t ldt, const float* __restrict__ scale, float* __restrict__ amax_acc) {
    __shared__ __attribute__((aligned(16))) unsigned tile[128][33];
    __shared__ float sh[4];
    const float inv = 1.f / scale[0];
    const float fmax_ = E5M2 ? 57344.f : 448.f;
    const int64_t r0 = (int64_t)blockIdx.y * 128, c0 = (int64_t)blockIdx.x * 128;
    const int ch = threadIdx.x & 7, rr = threadIdx.x >> 3;
    float am = 0.f;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int r = pass * 32 + rr;
        const bf16_t* src = x + (r0 + r) * ld + c0 + 16 * ch;
        const bf16x8 h0 = *reinterpret_cast<const bf16x8*>(src), h1 = *reinterpret_cast<const bf16x8*>(src + 8);
        unsigned w[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const bf16x8& h = g < 2 ? h0 : h1;
            const int b = 4 * (g & 1);
            f32x4 v = {(float)h[b], (float)h[b + 1], (float)h[b + 2], (float)h[b + 3]};
            if (amax_acc) {
                am = fmaxf(am, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
                v = v * inv;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) v[jj] = __builtin_amdgcn_fmed3f(v[jj], -fmax_, fmax_);
            } else {
                v = v * inv;
            }
            w[g] = fp8_pack4<E5M2>(v);
            tile[r][4 * ch + g] = w[g];
        }
        *reinterpret_cast<uint4*>(q + (r0 + r) * ldq + c0 + 16 * ch) = uint4{w[0], w[1], w[2], w[3]};
    }
    if (amax_acc) {
        am = wave_max(am);
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = am;
    }
    __syncthreads();
    if (amax_acc && threadIdx.x == 0) {
        const float m = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
        unsigned* acc = reinterpret_cast<unsigned*>(amax_acc);
        const unsigned mb = __float_as_uint(m);
        if (m > 0.f && !(m != m) && mb > __hip_atomic_load(acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(acc, mb);
    }
    fp8_tile128_transpose_store(tile, qt, ldt, r0, c0);
}

extern "C" int vaw_fp8_transpose(const void* q, int64_t R, int64_t C, int64_t ldq, void* qt, int64_t ldt, vaw_stream stream) {
    VAW_CHECK_ARG(q && qt && R > 0 && C > 0 && R % 64 == 0 && C % 128 == 0 && ldq >= C && ldq % 8 == 0 && ldt >= R && ldt % 4 == 0,
                  "fp8_transpose: R %% 64, C %% 128, ldq %% 8, ldt %% 4");
    VAW_CHECK_ARG((((uintptr_t)q) & 7) == 0 && (((uintptr_t)qt) & 3) == 0, "fp8_transpose: alignment");
    if (R % 128 == 0 && ldq % 16 == 0 && ldt % 16 == 0 && ((((uintptr_t)q) | ((uintptr_t)qt)) & 15) == 0) {
        dim3 grid128((unsigned)(C / 128), (unsigned)(R / 128));
        fp8_transpose128_kernel<<<grid128, 256, 0, (hipStream_t)stream>>>((const unsigned char*)q, ldq, (unsigned char*)qt, ldt);
        VAW_CHECK_LAUNCH("fp8_transpose");
        return VAW_OK;
    }
    dim3 grid((unsigned)(C / 128), (unsigned)(R / 64));
    fp8_transpose_kernel<<<grid, 256, 0, (hipStream_t)stream>>>((const unsigned char*)q, ldq, (unsigned char*)qt, ldt);
    VAW_CHECK_LAUNCH("fp8_transpose");
    return VAW_OK;
}

// Delayed scaling, once per step for all tensors: state = {scale in use, running max |x|, FMAX / margin, unused}.
__global__ void fp8_scale_update_kernel(float* __restrict__ states, int64_t n) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    float* st = states + 4 * i;
    const float m = st[1];
    if (m > 0.f) st[0] = m / st[2];      // an all-zero (or never quantised) tensor keeps its scale
    st[1] = 0.f;
}

extern "C" int64_t vaw_fp8_quantize_workspace_floats(void) { return 1024; }

extern "C" int vaw_fp8_quantize(vaw_dtype src_dt, vaw_dtype dst_format, const void* src, int64_t R, int64_t C, int64_t ld, void* q,
                                int64_t ldq, void* qt, int64_t ldt, float* scale_out, float* workspace, int64_t workspace_floats,
                                vaw_stream stream) {
    VAW_CHECK_ARG(src && q && scale_out && R > 0 && C > 0 && C % 4 == 0 && ld >= C && ld % 4 == 0 && ldq >= C && ldq % 4 == 0,
                  "fp8_quantize: sizes (C, ld, ldq multiples of 4)");
    VAW_CHECK_ARG(!qt || (ldt >= R && ldt % 4 == 0), "fp8_quantize: transposed copy needs ldt >= R, ldt %% 4 == 0");
    VAW_CHECK_ARG(src_dt == VAW_F32 || src_dt == VAW_BF16, "fp8_quantize: source must be f32 or bf16");
    VAW_CHECK_ARG(dst_format == VAW_FP8 || dst_format == VAW_BF8, "fp8_quantize: destination format VAW_FP8 (e4m3) or VAW_BF8 (e5m2)");
    VAW_CHECK_ARG(workspace && workspace_floats >= 1024, "fp8_quantize: workspace of vaw_fp8_quantize_workspace_floats() floats");
    VAW_CHECK_ARG(((uintptr_t)src & (src_dt == VAW_F32 ? 15 : 7)) == 0 && (((uintptr_t)q | (uintptr_t)qt) & 3) == 0, "fp8_quantize: alignment");
    hipStream_t s = (hipStream_t)stream;
    int64_t nb = (R * (C / 4) + 1023) / 1024;
    if (nb > 1024) nb = 1024;
    dim3 grid((unsigned)((C + 63) / 64), (unsigned)((R + 63) / 64));
    const bool e5 = dst_format == VAW_BF8;
    const float fmt_max = e5 ? 57344.f : 448.f;
    unsigned char *qp = (unsigned char*)q, *qtp = (unsigned char*)qt;
    const bool tiled = src_dt == VAW_BF16 && qt && R % 64 == 0 && C % 128 == 0 && ld % 8 == 0 && ldq % 8 == 0 &&
                       ((((uintptr_t)src) & 15) == 0) && ((((uintptr_t)q) & 7) == 0);
    if (tiled) {
        fp8_amax_partial_kernel<bf16_t><<<(int)nb, 256, 0, s>>>((const bf16_t*)src, R, C, ld, workspace);
        fp8_amax_final_kernel<<<1, 64, 0, s>>>(workspace, (int)nb, scale_out, fmt_max);
        const bool t128 = R % 128 == 0 && ld % 8 == 0 && ldq % 16 == 0 && ldt % 16 == 0 && ((((uintptr_t)q) | ((uintptr_t)qt)) & 15) == 0;
        dim3 gt((unsigned)(C / 128), (unsigned)(R / (t128 ? 128 : 64)));
        if (t128 && e5) fp8_quantize_bf16_tile128_kernel<true><<<gt, 256, 0, s>>>((const bf16_t*)src, ld, qp, ldq, qtp, ldt, scale_out, nullptr);
        else if (t128) fp8_quantize_bf16_tile128_kernel<false><<<gt, 256, 0, s>>>((const bf16_t*)src, ld, qp, ldq, qtp, ldt, scale_out, nullptr);
        else if (e5) fp8_quantize_bf16_tile_kernel<true><<<gt, 256, 0, s>>>((const bf16_t*)src, ld, qp, ldq, qtp, ldt, scale_out, nullptr);
        else fp8_quantize_bf16_tile_kernel<false><<<gt, 256, 0, s>>>((const bf16_t*)src, ld, qp, ldq, qtp, ldt, scale_out, nullptr);
        VAW_CHECK_LAUNCH("fp8_quantize");
        return VAW_OK;
    }
#define QUANT_GO(T, E5)                                                                                          \
    do {                                                                                                         \
        fp8_amax_partial_kernel<T><<<(int)nb, 256, 0, s>>>((const T*)src, R, C, ld, workspace);                  \
        fp8_amax_final_kernel<<<1, 64, 0, s>>>(workspace, (int)nb, scale_out, fmt_max);                          \
        fp8_quantize_kernel<T, E5><<<grid, 256, 0, s>>>((const T*)src, R, C, ld, qp, ldq, qtp, ldt, scale_out, nullptr);  \
    } while (0)
    if (src_dt == VAW_F32) { if (e5) QUANT_GO(float, true); else QUANT_GO(float, false); }
    else { if (e5) QUANT_GO(bf16_t, true); else QUANT_GO(bf16_t, false); }
    VAW_CHECK_LAUNCH("fp8_quantize");
    return VAW_OK;
}

extern "C" int vaw_fp8_quantize_delayed(vaw_dtype src_dt, vaw_dtype dst_format, const void* src, int64_t R, int64_t C, int64_t ld, void* q,
                                        int64_t ldq, void* qt, int64_t ldt, float* state, vaw_stream stream) {
    VAW_CHECK_ARG(src && q && state && R > 0 && C > 0 && C % 4 == 0 && ld >= C && ld % 4 == 0 && ldq >= C && ldq % 4 == 0,
                  "fp8_quantize_delayed: sizes (C, ld, ldq multiples of 4)");
    VAW_CHECK_ARG(!qt || (ldt >= R && ldt % 4 == 0), "fp8_quantize_delayed: transposed copy needs ldt >= R, ldt %% 4 == 0");
    VAW_CHECK_ARG(src_dt == VAW_F32 || src_dt == VAW_BF16, "fp8_quantize_delayed: source must be f32 or bf16");
    VAW_CHECK_ARG(dst_format == VAW_FP8 || dst_format == VAW_BF8, "fp8_quantize_delayed: destination format VAW_FP8 (e4m3) or VAW_BF8 (e5m2)");
    VAW_CHECK_ARG(((uintptr_t)src & (src_dt == VAW_F32 ? 15 : 7)) == 0 && (((uintptr_t)q | (uintptr_t)qt | (uintptr_t)state) & 3) == 0,
                  "fp8_quantize_delayed: alignment");
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((unsigned)((C + 63) / 64), (unsigned)((R + 63) / 64));
    const bool e5 = dst_format == VAW_BF8;
    unsigned char *qp = (unsigned char*)q, *qtp = (unsigned char*)qt;
    const bool tiled = src_dt == VAW_BF16 && qt && R % 64 == 0 && C % 128 == 0 && ld % 8 == 0 && ldq % 8 == 0 &&
                       ((((uintptr_t)src) & 15) == 0) && ((((uintptr_t)q) & 7) == 0);
    if (tiled) {
        const bool t128 = R % 128 == 0 && ld % 8 == 0 && ldq % 16 == 0 && ldt % 16 == 0 && ((((uintptr_t)q) | ((uintptr_t)qt)) & 15) == 0;
        dim3 gt((unsigned)(C / 128), (unsigned)(R / (t128 ? 128 : 64)));
        if (t128 && e5) fp8_quantize_bf16_tile128_kernel<true><<<gt, 256, 0, s>>>((const bf16_t*)src, ld, qp, ldq, qtp, ldt, state, state + 1);
        else if (t128) fp8_quantize_bf16_tile128_kernel<false><<<gt, 256, 0, s>>>((const bf16_t*)src, ld, qp, ldq, qtp, ldt, state, state + 1);
        else if (e5) fp8_quantize_bf16_tile_kernel<true><<<gt, 256, 0, s>>>((const bf16_t*)src, ld, qp, ldq, qtp, ldt, state, state + 1);
        else fp8_quantize_bf16_tile_kernel<false><<<gt, 256, 0, s>>>((const bf16_t*)src, ld, qp, ldq, qtp, ldt, state, state + 1);
        VAW_CHECK_LAUNCH("fp8_quantize_delayed");
        return VAW_OK;
    }
#define QUANT_D(T, E5) fp8_quantize_kernel<T, E5><<<grid, 256, 0, s>>>((const T*)src, R, C, ld, qp, ldq, qtp, ldt, state, state + 1)
    if (src_dt == VAW_F32) { if (e5) QUANT_D(float, true); else QUANT_D(float, false); }
    else { if (e5) QUANT_D(bf16_t, true); else QUANT_D(bf16_t, false); }
    VAW_CHECK_LAUNCH("fp8_quantize_delayed");
    return VAW_OK;
}

extern "C" int64_t vaw_fp8_quantize_batched_desc_bytes(int n_jobs) { return (int64_t)n_jobs * (int64_t)sizeof(QuantJobDev); }

extern "C" int vaw_fp8_quantize_delayed_batched(int n_jobs, const vaw_fp8_quant_job* jobs, void* desc_dev, int upload, vaw_stream stream) {
    VAW_CHECK_ARG(n_jobs > 0 && n_jobs <= 4096 && jobs && desc_dev, "fp8_quantize_delayed_batched: bad arguments");
    static thread_local QuantJobDev host[4096];
    int64_t blocks = 0;
    for (int j = 0; j < n_jobs; ++j) {
        const vaw_fp8_quant_job& q = jobs[j];
        VAW_CHECK_ARG(q.src && q.q && q.state && q.R > 0 && q.C > 0 && q.C % 4 == 0 && q.ld >= q.C && q.ld % 4 == 0 && q.ldq >= q.C && q.ldq % 4 == 0 &&
                      (!q.qt || (q.ldt >= q.R && q.ldt % 4 == 0)), "fp8_quantize_delayed_batched: job %d: sizes (C, ld, ldq, ldt multiples of 4)", j);
        VAW_CHECK_ARG(((uintptr_t)q.src & 15) == 0 && (((uintptr_t)q.q | (uintptr_t)q.qt | (uintptr_t)q.state) & 3) == 0, "fp8_quantize_delayed_batched: job %d: alignment", j);
        const int tx = (int)((q.C + 63) / 64), ty = (int)((q.R + 63) / 64);
        host[j] = QuantJobDev{q.src, (unsigned char*)q.q, (unsigned char*)q.qt, q.state, q.R, q.C, q.ld, q.ldq, q.ldt, (int)blocks, tx};
        blocks += (int64_t)tx * ty;
        VAW_CHECK_ARG(blocks < (1 << 30), "fp8_quantize_delayed_batched: too many tiles");
    }
    hipStream_t s = (hipStream_t)stream;
    if (upload) {
        const hipError_t rc = vaw_upload_table(desc_dev, host, sizeof(QuantJobDev) * n_jobs, s);
        VAW_CHECK_ARG(rc == hipSuccess, "fp8_quantize_delayed_batched: descriptor upload failed: %s", hipGetErrorString(rc));
    }
    fp8_quantize_batched_kernel<<<(unsigned)blocks, 256, 0, s>>>((const QuantJobDev*)desc_dev, n_jobs);
    VAW_CHECK_LAUNCH("fp8_quantize_delayed_batched");
    return VAW_OK;
}

extern "C" int vaw_fp8_scale_update(float* states, int64_t n, vaw_stream stream) {
    VAW_CHECK_ARG(states && n > 0 && ((uintptr_t)states & 15) == 0, "fp8_scale_update: states [n][4] floats, 16-byte aligned");
    fp8_scale_update_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(states, n);
    VAW_CHECK_LAUNCH("fp8_scale_update");
    return VAW_OK;
}

// ---- GEMM ---------------------------------------------------------------------------------------------------------------
extern "C" int vaw_gemm_fp8(vaw_dtype a_format, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const float* scale_a, const void* B, int64_t ldb,
                            const float* scale_b, void* C, int64_t ldc, const vaw_epilogue* ep, float* c_fp8_state, vaw_dtype c_fp8_format,
                            float* workspace, int64_t workspace_floats, vaw_stream stream) {
    VAW_CHECK_ARG(a_format == VAW_FP8 || a_format == VAW_BF8, "gemm_fp8: a_format VAW_FP8 (e4m3) or VAW_BF8 (e5m2)");
    VAW_CHECK_ARG(M >= 16 && N >= 16 && K > 0 && K % 128 == 0 && A && B && C, "gemm_fp8: sizes (K %% 128 == 0)");
    VAW_CHECK_ARG(lda >= K && ldb >= K && ldc >= N && lda % 16 == 0 && ldb % 16 == 0 && N % 8 == 0 && ldc % 8 == 0, "gemm_fp8: leading dimensions");
    VAW_CHECK_ARG((((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) & 15) == 0, "gemm_fp8: 16-byte alignment");
    EpiDev e{};
    e.alpha = 1.f;
    if (ep) {
        e.bias = ep->bias; e.act = ep->act; e.aux_in = ep->aux_in; e.aux_out = ep->aux_out; e.gate = ep->gate;
        e.gate_ld = ep->gate_ld; e.resid = ep->resid; e.rowadd = ep->rowadd; e.rpb = ep->rows_per_batch;
        e.alpha = ep->alpha; e.beta = ep->beta; e.out_f32 = ep->out_f32; e.resid_act = ep->resid_is_act;
    }
    VAW_CHECK_ARG(!(ep && ep->rowsum_a_out), "gemm_fp8: rowsum_a_out is not offered (take bias gradients from the bf16 tensors)");
    VAW_CHECK_ARG(e.act >= 0 && e.act <= 2 && (e.act != 2 || e.aux_in), "gemm_fp8: act");
    VAW_CHECK_ARG(!(e.gate || e.rowadd) || e.rpb > 0, "gemm_fp8: gate/rowadd need rows_per_batch");
    VAW_CHECK_ARG(!(e.act == 2 && e.gate) && !(e.resid && e.rowadd), "gemm_fp8: epilogue combination not offered");
    if (e.rpb <= 0) e.rpb = 1;
    float* const colsum_final = ep ? ep->colsum_out : nullptr;
    float* const colsum_part = ep ? ep->colsum_partial_out : nullptr;       // deferred fold (vaw_reduce_rows_batched): see vaw_epilogue
    VAW_CHECK_ARG(!colsum_part || (!colsum_final && ep->colsum_rows_out), "gemm_fp8: colsum_partial_out excludes colsum_out and needs colsum_rows_out");
    const bool colsum_out = colsum_final || colsum_part;
    VAW_CHECK_ARG(!colsum_part || *ep->colsum_rows_out >= (M + 127) / 128, "gemm_fp8: colsum_partial_out holds %ld rows, this launch writes %ld",
                  (long)*ep->colsum_rows_out, (long)((M + 127) / 128));      // colsum_rows_out: capacity in, rows written out
    VAW_CHECK_ARG(!colsum_final || (workspace && workspace_floats >= ((M + 127) / 128) * N), "gemm_fp8: colsum_out needs a workspace");
    e.M = M; e.N = N; e.ldc = ldc; e.C = C; e.slab = workspace; e.nt_off = 1;
    e.scale_a = scale_a; e.scale_b = scale_b;
    {
        static int dbg = -1;
        if (dbg < 0) { const char* v = getenv("VAW_GEMM_DEBUG"); dbg = v ? atoi(v) : 0; }
        e.debug = dbg;
    }
    e.colpart = colsum_part ? colsum_part : colsum_final ? workspace : nullptr;
    // plan in units of the kernel's K tiles: 128 fp8 elements = one K tile = what 64 bf16 elements are to vaw_p8_plan
    // (plain_f32 = false: no K split -- the long-K launches of the step are the weight gradients, served by vaw_wgrad_grouped)
    const P8Plan pl = vaw_p8_plan(M, N, K / 2, false, colsum_out, workspace_floats, 1);
    hipStream_t s = (hipStream_t)stream;
    const int bn = 64 * pl.ntw, tiles_m = (int)((M + 255) / 256), tiles_n = (int)((N + bn - 1) / bn), nk = (int)(K / 128);
    int epi;
    const bool bf16_out = !e.out_f32;
    e.q_state = c_fp8_state;
    e.q_e5m2 = c_fp8_format == VAW_BF8;
    VAW_CHECK_ARG(!c_fp8_state || ((c_fp8_format == VAW_FP8 || c_fp8_format == VAW_BF8) && bf16_out && (e.act == 1 || e.act == 2) && ldc % 8 == 0),
                  "gemm_fp8: fp8 output is offered for the GELU / GELU' epilogues (C then holds bytes, row stride ldc bytes)");
    if (e.act == 1 && e.aux_out && !e.gate && !e.resid && !e.rowadd && bf16_out && !e.colpart) epi = c_fp8_state ? P8_GELU_Q : P8_GELU;
    else if (e.act == 2 && !e.bias && !e.aux_out && !e.gate && !e.resid && !e.rowadd && bf16_out) epi = c_fp8_state ? P8_DGELU_Q : P8_DGELU;
    else if (e.act == 0 && e.gate && e.resid && !e.resid_act && e.aux_out && !e.rowadd && e.out_f32 && e.beta == 0.f && !e.colpart) epi = P8_GATE;
    else if (e.act == 0 && !e.aux_out && !e.gate && !e.resid && !e.rowadd && e.beta == 0.f) epi = P8_STORE;
    else {
        vaw_set_error("gemm_fp8: this epilogue combination has no fp8 kernel");
        return VAW_ERR_UNSUPPORTED;
    }
    // instantiated: what the training step launches -- forward kinds with e4m3 activations, input-gradient kinds with e5m2 or
    // e4m3 gradients
#define F8_GO(EPIv, FMT)                                                                                                         \
    do {                                                                                                                         \
        if (pl.ntw == 4) p8_launch_fp8_one<4, EPIv, FMT>(A, lda, B, ldb, nk, tiles_m, tiles_n, pl.split, pl.grid, e, s);          \
        else p8_launch_fp8_one<3, EPIv, FMT>(A, lda, B, ldb, nk, tiles_m, tiles_n, pl.split, pl.grid, e, s);                      \
    } while (0)
    const bool e5 = a_format == VAW_BF8;
    if (epi == P8_STORE) { if (e5) F8_GO(P8_STORE, 2); else F8_GO(P8_STORE, 1); }
    else if (epi == P8_DGELU) { if (e5) F8_GO(P8_DGELU, 2); else F8_GO(P8_DGELU, 1); }
    else if (epi == P8_DGELU_Q) { if (e5) F8_GO(P8_DGELU_Q, 2); else F8_GO(P8_DGELU_Q, 1); }
    else if (e5) {
        vaw_set_error("gemm_fp8: the forward epilogues (GELU, gated residual) take e4m3 activations");
        return VAW_ERR_UNSUPPORTED;
    } else if (epi == P8_GELU) F8_GO(P8_GELU, 1);
    else if (epi == P8_GELU_Q) F8_GO(P8_GELU_Q, 1);
    else F8_GO(P8_GATE, 1);
    VAW_CHECK_LAUNCH("gemm_fp8");
    if (colsum_part) *ep->colsum_rows_out = (M + 127) / 128;
    else if (colsum_final) return vaw_reduce_rows(workspace, (M + 127) / 128, N, colsum_final, ep->colsum_beta, stream);
    return VAW_OK;
}
