// Epilogue descriptor and the row-wise epilogue arithmetic shared by the MFMA GEMM kernels (gemm.hip, gemm_p8.hip).
#pragma once
#include "common.h"

struct EpiDev {
    const float* bias;
    int act;
    const void* aux_in;
    void* aux_out;
    const float* gate;
    int64_t gate_ld;
    const void* resid;
    int resid_act;    // 1: resid has act dtype (UNet skip/residual adds), 0: f32 (DiT residual stream)
    const float* rowadd;
    int rpb;
    float alpha, beta;
    int out_f32;
    int64_t M, N, ldc;
    void* C;
    float* slab;   // split-K partial sums [n_split][M][N] f32 (workspace), or NULL
    int debug;        // measurement only (VAW_GEMM_DEBUG): 1 = skip the epilogue, 2 = skip the K loop (ablations 3-5 of DESIGN.md §5 lived here)
    int direct_epi;   // 1: register-direct epilogue (default), 0: LDS-staged (VAW_GEMM_EPI=0; always for fused column sums)
    float* rowpart;   // mn-major A only (CONV 3 / plain weight gradients): [n_split][M] f32 partial row sums of A = dy^T
                      // over this split's K range (the layer's bias gradient)
    float* colpart;   // [M/128][N] f32: per-row-tile column sums of the OUTPUT (bias gradient of the next layer), or NULL
};

// Streaming (non-temporal) 16-byte stores for the epilogue: the output tile is written once and not re-read by this
// launch, so it should not evict the operand panels other workgroups of the XCD are re-reading from L2.
__device__ __forceinline__ void nt_store(float* p, f32x4 v) { __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p)); }
__device__ __forceinline__ void nt_store(bf16_t* p, bf16x8 v) { __builtin_nontemporal_store(v, reinterpret_cast<bf16x8*>(p)); }

// One accumulator fragment row: 4 consecutive rows (m..m+3) at one column n.
template <typename TO>
__device__ __forceinline__ void epi_store4(const EpiDev& e, int64_t m, int64_t n, f32x4 acc) {
    if (n >= e.N) return;
    const float bias = e.bias ? e.bias[n] : 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t mm = m + j;
        if (mm >= e.M) break;
        const int64_t off = mm * e.ldc + n;
        float v = acc[j] * e.alpha + bias;
        if (e.aux_out) {
            TO r = from_f32<TO>(v);
            ((TO*)e.aux_out)[off] = r;
            v = to_f32(r);
        }
        if (e.act == 1) v = gelu_tanh_f(v);
        else if (e.act == 2) v *= gelu_tanh_grad_f(to_f32(((const TO*)e.aux_in)[off]));
        const unsigned mu = (unsigned)mm, rpb = (unsigned)e.rpb;
        if (e.gate) v *= e.gate[(int64_t)(mu / rpb) * e.gate_ld + n];
        if (e.resid) v += e.resid_act ? to_f32(((const TO*)e.resid)[off]) : ((const float*)e.resid)[off];
        if (e.rowadd) v += e.rowadd[(int64_t)(mu % rpb) * e.N + n];
        if (e.out_f32) {
            float* c = (float*)e.C + off;
            *c = (e.beta != 0.f ? e.beta * *c : 0.f) + v;
        } else {
            ((TO*)e.C)[off] = from_f32<TO>(v);
        }
    }
}

// Eight consecutive columns n..n+7 of row m, all operands 16-byte aligned (fast path, second epilogue phase).
__device__ __forceinline__ void epi_row8(const EpiDev& e, unsigned m, int64_t n, f32x4& v0, f32x4& v1, f32x4 b0, f32x4 b1) {
    const int64_t off = (int64_t)m * e.ldc + n;
    v0 = v0 * e.alpha + b0;
    v1 = v1 * e.alpha + b1;
    if (e.aux_out) {
        // the saved branch value is the bf16-rounded one, and so is what the activation sees (fwd/bwd consistent)
        const bf16x8 r = {(bf16_t)v0[0], (bf16_t)v0[1], (bf16_t)v0[2], (bf16_t)v0[3],
                          (bf16_t)v1[0], (bf16_t)v1[1], (bf16_t)v1[2], (bf16_t)v1[3]};
        nt_store((bf16_t*)e.aux_out + off, r);
        v0 = f32x4{(float)r[0], (float)r[1], (float)r[2], (float)r[3]};
        v1 = f32x4{(float)r[4], (float)r[5], (float)r[6], (float)r[7]};
    }
    if (e.act == 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { v0[j] = gelu_tanh_f(v0[j]); v1[j] = gelu_tanh_f(v1[j]); }
    } else if (e.act == 2) {
        const bf16_t* ai = (const bf16_t*)e.aux_in + off;
        f32x4 h0 = load4(ai), h1 = load4(ai + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { v0[j] *= gelu_tanh_grad_f(h0[j]); v1[j] *= gelu_tanh_grad_f(h1[j]); }
    }
    const unsigned rpb = (unsigned)e.rpb;
    if (e.gate) {
        const float* g = e.gate + (int64_t)(m / rpb) * e.gate_ld + n;
        v0 *= load4(g);
        v1 *= load4(g + 4);
    }
    if (e.resid) {
        if (e.resid_act) {
            v0 += load4((const bf16_t*)e.resid + off);
            v1 += load4((const bf16_t*)e.resid + off + 4);
        } else {
            v0 += load4((const float*)e.resid + off);
            v1 += load4((const float*)e.resid + off + 4);
        }
    }
    if (e.rowadd) {
        const float* ra = e.rowadd + (int64_t)(m % rpb) * e.N + n;
        v0 += load4(ra);
        v1 += load4(ra + 4);
    }
    if (e.out_f32) {
        float* c = (float*)e.C + off;
        if (e.beta != 0.f) {
            v0 += e.beta * load4(c);
            v1 += e.beta * load4(c + 4);
        }
        nt_store(c, v0);
        nt_store(c + 4, v1);
    } else {
        bf16_t* c = (bf16_t*)e.C + off;
        bf16x8 r = {(bf16_t)v0[0], (bf16_t)v0[1], (bf16_t)v0[2], (bf16_t)v0[3], (bf16_t)v1[0], (bf16_t)v1[1], (bf16_t)v1[2], (bf16_t)v1[3]};
        nt_store(c, r);
        v0 = f32x4{(float)r[0], (float)r[1], (float)r[2], (float)r[3]};   // what a later reader of C sees
        v1 = f32x4{(float)r[4], (float)r[5], (float)r[6], (float)r[7]};
    }
}

// Four consecutive columns n..n+3 of row m straight from a TRANSPOSED accumulator tile (the MFMA is issued with its
// operands swapped, so a lane holds 4 consecutive columns of one row): the direct epilogue, no LDS round trip.
__device__ __forceinline__ void epi_row4(const EpiDev& e, unsigned m, int64_t n, f32x4 v, f32x4 b) {
    const int64_t off = (int64_t)m * e.ldc + n;
    v = v * e.alpha + b;
    if (e.aux_out) {
        const bf16x4 r = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
        __builtin_nontemporal_store(r, reinterpret_cast<bf16x4*>((bf16_t*)e.aux_out + off));
        v = f32x4{(float)r[0], (float)r[1], (float)r[2], (float)r[3]};
    }
    if (e.act == 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = gelu_tanh_f(v[j]);
    } else if (e.act == 2) {
        const f32x4 h = load4((const bf16_t*)e.aux_in + off);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] *= gelu_tanh_grad_f(h[j]);
    }
    const unsigned rpb = (unsigned)e.rpb;
    if (e.gate) v *= load4(e.gate + (int64_t)(m / rpb) * e.gate_ld + n);
    if (e.resid) v += e.resid_act ? load4((const bf16_t*)e.resid + off) : load4((const float*)e.resid + off);
    if (e.rowadd) v += load4(e.rowadd + (int64_t)(m % rpb) * e.N + n);
    if (e.out_f32) {
        float* c = (float*)e.C + off;
        if (e.beta != 0.f) v += e.beta * load4(c);
        nt_store(c, v);
    } else {
        const bf16x4 r = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
        __builtin_nontemporal_store(r, reinterpret_cast<bf16x4*>((bf16_t*)e.C + off));
    }
}

