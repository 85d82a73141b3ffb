// Shared device helpers of the bf16 MFMA attention kernels (attention_mfma.hip, attention_bwd_big.hip): LDS image layout and
// staging, fragment reads in both orientations, output staging, column sums.  See attention_mfma.hip for the scheme.
#pragma once
#include "common.h"

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

struct AttnMfmaArgs {
    int B, H, T;
    int64_t q_sb, q_sh, q_st;   // element strides of q/k/v (and dq/dk/dv); channel stride is 1
    int64_t o_sb, o_sh, o_st;   // element strides of o / d_o
    float scale;
    int hd;        // true head dim (multiple of 8, <= the template's HD): LDS columns hd..HD-1 are zero-filled
};
static __device__ __attribute__((aligned(64))) const unsigned char attn_zero_page[64] = {0};

template <int HD>
__device__ __forceinline__ int swz(int row) {
    return HD == 32 ? ((-(row >> 2)) & 3) : HD == 64 ? ((row >> 1) & 7) : HD == 128 ? (row & 15) : 0;
}
template <int HD> struct Img {
    static constexpr int PCH = HD == 96 ? 13 : HD / 8;     // 16-byte chunks per image row (pitch)
    static constexpr int PITCH = 16 * PCH;
    static constexpr int BYTES = 64 * PITCH;                // one 64-row image
};
template <int HD>
__device__ __forceinline__ int img_off(int row, int chunk) { return row * Img<HD>::PITCH + ((chunk ^ swz<HD>(row)) << 4); }

// 64 token rows starting at g (row stride stride_t elements) -> LDS image rows [0,64); all 4 waves cooperate
template <int HD>
__device__ __forceinline__ void stage_block(const bf16_t* __restrict__ g, int64_t stride_t, char* img, int wid, int lane,
                                            int hd) {
    constexpr int PCH = Img<HD>::PCH;      // 16-byte chunks per row incl. padding; the image is PCH wave-instructions of 1 KiB
#pragma unroll
    for (int i = 0; i < (PCH + 3) / 4; ++i) {
        const int inst = wid + 4 * i;
        if (PCH % 4 != 0 && inst >= PCH) break;              // uniform per wave
        const int idx = inst * 64 + lane;
        const int row = idx / PCH;
        const int chunk = (idx % PCH) ^ swz<HD>(row);
        const bf16_t* src = chunk * 8 < hd ? g + (int64_t)row * stride_t + chunk * 8
                                           : reinterpret_cast<const bf16_t*>(attn_zero_page);   // padded head dim (72 -> 96), pad chunk
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(img + inst * 1024), 16, 0, 0);
    }
}

// Register-staged variant of stage_block (cdna_hip_programming.md T14, "async-STAGE split"): blk_prefetch ISSUES the global
// loads of a 64-row block into registers (3 x 16 bytes per thread for the 96-wide images) and returns at once; blk_commit writes
// them into the LDS image later, after the barrier that frees it.  The kernels that stream key / query blocks issue the loads
// of block i + 1 before they compute on block i: with one image per operand (3 workgroups per CU stay resident) every block
// used to cost a full memory latency behind DMA_WAIT_SYNC -- a workgroup of the DiT-XL/2 forward was resident for 25 us to do
// 1.5 us of MFMAs.  Channels hd .. HD-1 are written as zeros; the pad chunk of the 96-wide pitch is never read.
template <int HD> struct BlkRegs {
    static constexpr int N = (64 * (HD / 8) + 255) / 256;          // 16-byte chunks per thread
    bf16x8 v[N];
};
template <int HD>
__device__ __forceinline__ void blk_prefetch(BlkRegs<HD>& r, const bf16_t* __restrict__ g, int64_t stride_t, int hd) {
    constexpr int CPR = HD / 8;
#pragma unroll
    for (int i = 0; i < BlkRegs<HD>::N; ++i) {
        const int idx = threadIdx.x + 256 * i;
        const int row = idx / CPR, chunk = idx - row * CPR;
        const bool live = (64 * CPR % 256 == 0 || idx < 64 * CPR) && chunk * 8 < hd;
        r.v[i] = live ? *reinterpret_cast<const bf16x8*>(g + (int64_t)row * stride_t + chunk * 8)
                      : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    }
}
template <int HD>
__device__ __forceinline__ void blk_commit(const BlkRegs<HD>& r, char* img) {
    constexpr int CPR = HD / 8;
#pragma unroll
    for (int i = 0; i < BlkRegs<HD>::N; ++i) {
        const int idx = threadIdx.x + 256 * i;
        const int row = idx / CPR, chunk = idx - row * CPR;
        if (64 * CPR % 256 == 0 || idx < 64 * CPR) *reinterpret_cast<bf16x8*>(img + img_off<HD>(row, chunk)) = r.v[i];
    }
}

// Output staging: gradients / outputs leave through LDS as whole rows.  A lane of an accumulator tile owns 4 bf16 of one row, so a
// direct store instruction writes 32-byte pieces of 16 rows; staged over a dead operand image, eight lanes write one 128-byte row
// with 16 bytes each (T = 64 backward: 53.8 -> 43.5 us).  Tile layout: [rows][2 HD bytes] (HD 96: the images' 208-byte pitch), 16-byte
// chunk c of row r at c ^ (r & 7).
template <int HD>
__device__ __forceinline__ int out_off(int r, int c16) {
    return HD == 96 ? r * 208 + 16 * c16 : r * (2 * HD) + ((c16 ^ (r & (HD / 8 < 8 ? HD / 8 - 1 : 7))) << 4);
}
// this wave's 16 rows r16 .. r16 + 15 of a staged tile: lane (li, g) -> row r16 + li, columns 16 dt + 4 g ..
template <int HD, int DT>
__device__ __forceinline__ void out_stage16(char* dst, int r16, const f32x4 (&acc)[DT], int lane) {
    const int r = r16 + (lane & 15), g = lane >> 4;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
        const bf16x4 w = {(bf16_t)acc[dt][0], (bf16_t)acc[dt][1], (bf16_t)acc[dt][2], (bf16_t)acc[dt][3]};
        *reinterpret_cast<bf16x4*>(dst + out_off<HD>(r, 2 * dt + (g >> 1)) + 8 * (g & 1)) = w;
    }
}
// all 256 threads: `rows` staged rows -> global rows g + r * stride, 16 bytes per access, channels < hd only
template <int HD>
__device__ __forceinline__ void out_flush(const char* src, int rows, bf16_t* __restrict__ g, int64_t stride, int hd) {
    constexpr int CPR = HD / 8;
    for (int idx = threadIdx.x; idx < rows * CPR; idx += 256) {
        const int r = idx / CPR, c16 = idx - r * CPR;
        if (8 * c16 < hd) *reinterpret_cast<bf16x8*>(g + (int64_t)r * stride + 8 * c16) = *reinterpret_cast<const bf16x8*>(src + out_off<HD>(r, c16));
    }
}

// 16 rows x 32 k (k = channel), rows r0.., k-step s: the natural A (or B) fragment
template <int HD>
__device__ __forceinline__ bf16x8 frag_rows(const char* img, int r0, int s, int lane) {
    return *reinterpret_cast<const bf16x8*>(img + img_off<HD>(r0 + (lane & 15), 4 * s + (lane >> 4)));
}
// Transposed fragment: operand row = image column d0 + (l&15), k = image rows in the accumulator-derived order
// {kbase + 4g + 0..3, kbase + 16 + 4g + 0..3}, g = l>>4.
template <int HD>
__device__ __forceinline__ bf16x8 frag_cols_perm(const char* img, int d0, int kbase, int lane) {
    const int li = lane & 15, q = li >> 2, p = li & 3, g = lane >> 4;
    const int ch = (d0 >> 3) + (p >> 1);
    const int r_lo = kbase + 4 * g + q, r_hi = r_lo + 16;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(img + img_off<HD>(r_lo, ch) + 8 * (p & 1)));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(img + img_off<HD>(r_hi, ch) + 8 * (p & 1)));
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

// Column sums of a [rows][hd] gradient block held as transposed accumulator tiles (a lane owns row li = lane & 15 of its wave's
// 16 rows, columns 16 dt + 4 g .. + 3): the values AS STORED (bf16 roundings) are summed over the 16 rows of the wave by
// shuffles and left in this wave's slot cs_w [16 DT] of the workgroup's LDS scratch; attn_cs_commit folds the four waves in a
// fixed order.  The qkv bias gradient (autograd of timm Attention's qkv Linear, models/dit.py:126) is the sum of these over all
// tokens: taking them here saves re-reading dqkv (75 MB per DiT-B/4 block) in a separate column-sum pass.
__device__ __forceinline__ f32x4 as_stored_bf16(f32x4 a) {
    return f32x4{(float)(bf16_t)a[0], (float)(bf16_t)a[1], (float)(bf16_t)a[2], (float)(bf16_t)a[3]};
}
template <int DT>
__device__ __forceinline__ void attn_cs_wave(const f32x4 (&stored)[DT], float* cs_w, int lane) {      // stored: as_stored_bf16 values
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
        f32x4 v = stored[dt];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[j] += __shfl_xor(v[j], 1, 64); v[j] += __shfl_xor(v[j], 2, 64);
            v[j] += __shfl_xor(v[j], 4, 64); v[j] += __shfl_xor(v[j], 8, 64);
        }
        if ((lane & 15) == 0) store4(cs_w + 16 * dt + 4 * (lane >> 4), v);
    }
}
// after a workgroup barrier: out[c] = ((w0 + w1) + w2) + w3 for the hd real columns of `n_which` quantities (cs: [which][4 waves][HD])
// quantity w of the scratch goes to columns (first_which + w) * H hd + h hd + c of the partial row (packed qkv order [3][H][hd])
template <int HD>
__device__ __forceinline__ void attn_cs_commit(const float* cs, int n_which, int first_which, int hd, int Hhd, int h, float* out_row) {
    for (int i = threadIdx.x; i < n_which * HD; i += blockDim.x) {
        const int w = i / HD, c = i - w * HD;
        if (c >= hd) continue;
        const float* p = cs + w * 4 * HD + c;
        out_row[(first_which + w) * Hhd + h * hd + c] = ((p[0] + p[HD]) + p[2 * HD]) + p[3 * HD];
    }
}
__device__ __forceinline__ float group_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    v = fmaxf(v, __shfl_xor(v, 32, 64));
    return v;
}
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#define DMA_WAIT_SYNC()                                 \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    \
    __syncthreads()

