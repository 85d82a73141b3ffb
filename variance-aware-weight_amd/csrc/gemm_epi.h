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
    const float* scale_a;   // fp8 operands: device scalars (per-tensor dequantisation scales) multiplied into alpha, or NULL
    const float* scale_b;
    int nt_off;       // 1 (default): gemm_p8 epilogue stores use the default cache policy; 0 (VAW_P8_NT=1): non-temporal
    int nt_aux;       // 1 (default; VAW_P8_NT_AUX=0 switches off): the aux_out stores alone are non-temporal (a tensor only the backward pass re-reads)
    int debug;        // measurement only (VAW_GEMM_DEBUG): 1 = skip the epilogue, 2 = skip the K loop (ablations 3-5 of DESIGN.md §5 lived here)
    int direct_epi;   // 1: register-direct epilogue (default), 0: LDS-staged (VAW_GEMM_EPI=0; always for fused column sums)
    float* rowpart;   // mn-major A only (CONV 3 / plain weight gradients): [n_split][M] f32 partial row sums of A = dy^T
                      // over this split's K range (the layer's bias gradient)
    float* colpart;   // [M/128][N] f32: per-row-tile column sums of the OUTPUT (bias gradient of the next layer), or NULL
    float* q_state;   // fp8 OUTPUT (P8_GELU_Q / P8_DGELU_Q): the tensor's delayed-scaling state {scale in use, running max |x|}
    int q_e5m2;       // its format: 0 e4m3, 1 e5m2
    float q_inv;      // filled in by the kernel: 1 / q_state[0]
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
// (v comes back as the value a later reader of C sees: callers that also carry column sums add it up)
__device__ __forceinline__ void epi_row4(const EpiDev& e, unsigned m, int64_t n, f32x4& v, f32x4 b) {
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
        v = f32x4{(float)r[0], (float)r[1], (float)r[2], (float)r[3]};
    }
}


// ---- the same row epilogue in two halves, specialised at compile time, for the persistent kernel (gemm_p8.hip), which
// software-pipelines it: epi_load8 issues the global loads of one 8-column group one step ahead of the stores of the
// previous group (the compiler may not hoist loads over possibly-aliasing stores itself), epi_apply8 does the arithmetic and
// the stores.  EPI names the launch kinds of the training step; P8_ANY keeps every switch at run time.  GELU here goes through
// v_rcp_f32 instead of the IEEE division sequence (bf16 MFMA path only; the f32 parity kernels keep gelu_tanh_f).
enum { P8_STORE = 0,   // value = acc*alpha (+ bias) -> C (act dtype or f32)                      qkv forward, plain input gradients
       P8_GELU = 1,    // (+ bias) -> aux_out (bf16) -> GELU -> C (bf16)                          fc1 forward
       P8_DGELU = 2,   // * GELU'(aux_in) -> C (bf16), column sums optional                       fc2 input gradient
       P8_GATE = 3,    // (+ bias) -> aux_out (bf16) -> * gate + resid (f32) -> C (f32)           proj / fc2 forward
       P8_SLAB = 4,    // raw f32 partial sums of one K split                                     weight gradients
       P8_ANY = 5,
       P8_WGRAD = 6,   // weight gradients: whole tiles C = beta * C + acc (f32), K-split tiles raw into their slab
       P8_RESID = 7,   // (+ bias) + resid (act dtype) -> C (bf16)                                  conv forward with a fused skip add
       P8_GELU_Q = 8,  // P8_GELU with C written as fp8 bytes (the bf16 rounding of the value, scaled by 1 / q_state[0], saturated),
       P8_DGELU_Q = 9 };// P8_DGELU likewise: fp8 mode's `a` and `dhid`, whose bf16 forms nobody else reads; running max |x| -> q_state[1]
template <int EPI> struct EpiKind {
    static __device__ __forceinline__ bool act1(const EpiDev& e) { return EPI == P8_GELU || EPI == P8_GELU_Q || (EPI == P8_ANY && e.act == 1); }
    static __device__ __forceinline__ bool act2(const EpiDev& e) { return EPI == P8_DGELU || EPI == P8_DGELU_Q || (EPI == P8_ANY && e.act == 2); }
    static __device__ __forceinline__ bool gate(const EpiDev& e) { return EPI == P8_GATE || (EPI == P8_ANY && e.gate != nullptr); }
    static __device__ __forceinline__ bool resid(const EpiDev& e) { return EPI == P8_GATE || EPI == P8_RESID || (EPI == P8_ANY && e.resid != nullptr); }
    static __device__ __forceinline__ bool resid_act(const EpiDev& e) { return EPI == P8_RESID || (EPI == P8_ANY && e.resid_act); }
    static __device__ __forceinline__ bool rowadd(const EpiDev& e) { return EPI == P8_ANY && e.rowadd != nullptr; }
    static __device__ __forceinline__ bool aux_out(const EpiDev& e) { return EPI == P8_GELU || EPI == P8_GELU_Q || EPI == P8_GATE || (EPI == P8_ANY && e.aux_out != nullptr); }
    static __device__ __forceinline__ bool out_f32(const EpiDev& e) { return EPI == P8_GATE || EPI == P8_WGRAD || ((EPI == P8_ANY || EPI == P8_STORE) && e.out_f32); }
    static __device__ __forceinline__ bool beta(const EpiDev& e) { return (EPI == P8_ANY || EPI == P8_WGRAD) && e.beta != 0.f; }
    static __device__ __forceinline__ bool colsum(const EpiDev& e) { return (EPI == P8_DGELU || EPI == P8_DGELU_Q || EPI == P8_STORE || EPI == P8_ANY) && e.colpart != nullptr; }
    static constexpr bool may_colsum = EPI == P8_DGELU || EPI == P8_DGELU_Q || EPI == P8_STORE || EPI == P8_ANY;   // the sums are then always carried
    static constexpr bool out_q = EPI == P8_GELU_Q || EPI == P8_DGELU_Q;                      // C is fp8 bytes
    static constexpr bool loads = EPI == P8_DGELU || EPI == P8_DGELU_Q || EPI == P8_GATE || EPI == P8_RESID || EPI == P8_ANY;     // epi_load8 has something to fetch
};
// Two operand slots: x = GELU'-argument (act == 2) or gate; y = residual or row-add (each pair is mutually exclusive in every
// launch of the training step; the dispatcher keeps launches that set both members of a pair on the other kernel).
struct EpiOps {
    f32x4 x0, x1, y0, y1;
};
__device__ __forceinline__ float gelu_tanh_fast(float x) {
    const float k0 = 0.7978845608028654f, k1 = 0.044715f;
    const float u2 = 2.f * k0 * (x + k1 * x * x * x);
    return x * __builtin_amdgcn_rcpf(1.f + __expf(-u2));
}
__device__ __forceinline__ float gelu_tanh_grad_fast(float x) {
    const float k0 = 0.7978845608028654f, k1 = 0.044715f;
    const float x2 = x * x;
    const float u2 = 2.f * k0 * (x + k1 * x * x2);
    const float s = __builtin_amdgcn_rcpf(1.f + __expf(-u2));
    return s + x * s * (1.f - s) * (2.f * k0 * (1.f + 3.f * k1 * x2));
}
// (sample, row-in-sample) of row m are passed in: callers that walk rows in steps carry them instead of dividing
template <int EPI>
__device__ __forceinline__ void epi_load8(const EpiDev& e, unsigned m, int64_t n, unsigned sample, unsigned row_in_sample, EpiOps& o) {
    using K = EpiKind<EPI>;
    const int64_t off = (int64_t)m * e.ldc + n;
    if (K::act2(e)) {
        const bf16x8 h = *reinterpret_cast<const bf16x8*>((const bf16_t*)e.aux_in + off);
        o.x0 = f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
        o.x1 = f32x4{(float)h[4], (float)h[5], (float)h[6], (float)h[7]};
    } else if (K::gate(e)) {
        const float* g = e.gate + (int64_t)sample * e.gate_ld + n;
        o.x0 = load4(g);
        o.x1 = load4(g + 4);
    }
    if (K::resid(e)) {
        if (K::resid_act(e)) {
            const bf16x8 r = *reinterpret_cast<const bf16x8*>((const bf16_t*)e.resid + off);
            o.y0 = f32x4{(float)r[0], (float)r[1], (float)r[2], (float)r[3]};
            o.y1 = f32x4{(float)r[4], (float)r[5], (float)r[6], (float)r[7]};
        } else {
            o.y0 = load4((const float*)e.resid + off);
            o.y1 = load4((const float*)e.resid + off + 4);
        }
    } else if (K::rowadd(e)) {
        const float* ra = e.rowadd + (int64_t)row_in_sample * e.N + n;
        o.y0 = load4(ra);
        o.y1 = load4(ra + 4);
    }
}
// Stores of the pipelined epilogue go through buffer resources based at the tile's first element: a lane beyond the matrix
// edge gets an out-of-range offset and the hardware drops its store.  No branch around the stores, so the compiler can
// count its s_waitcnt for the operand loads that are in flight across them (a store inside a conditional block forces
// vmcnt(0) at the join: the whole memory pipe drains once per step).
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
#define EPI_OOB 0xFFFFFFF0u
__device__ __forceinline__ __amdgpu_buffer_rsrc_t epi_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)0x80000000u, 0x00020000);
}
__device__ __forceinline__ void buf_store16(__amdgpu_buffer_rsrc_t rs, unsigned byte_off, f32x4 v, bool nt) {
    if (nt) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), rs, byte_off, 0, 2);
    else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), rs, byte_off, 0, 0);
}
__device__ __forceinline__ void buf_store16(__amdgpu_buffer_rsrc_t rs, unsigned byte_off, bf16x8 v, bool nt) {
    if (nt) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), rs, byte_off, 0, 2);
    else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), rs, byte_off, 0, 0);
}
// loc = element offset of (row, first column) from the tile's first element, or a negative value for "store nothing".
// The beta read (P8_ANY) uses the plain pointer c_row = address of that row group when valid.
template <int EPI>
__device__ __forceinline__ void epi_apply8(const EpiDev& e, __amdgpu_buffer_rsrc_t rs_c, __amdgpu_buffer_rsrc_t rs_aux, int loc,
                                           const float* c_f32, f32x4& v0, f32x4& v1, f32x4 b0, f32x4 b1, const EpiOps& o, float& qmax) {
    using K = EpiKind<EPI>;
    const bool ok = loc >= 0;
    v0 = v0 * e.alpha + b0;
    v1 = v1 * e.alpha + b1;
    if (K::aux_out(e)) {
        const bf16x8 r = {(bf16_t)v0[0], (bf16_t)v0[1], (bf16_t)v0[2], (bf16_t)v0[3],
                          (bf16_t)v1[0], (bf16_t)v1[1], (bf16_t)v1[2], (bf16_t)v1[3]};
        buf_store16(rs_aux, ok ? 2u * (unsigned)loc : EPI_OOB, r, !e.nt_off || e.nt_aux);
        v0 = f32x4{(float)r[0], (float)r[1], (float)r[2], (float)r[3]};
        v1 = f32x4{(float)r[4], (float)r[5], (float)r[6], (float)r[7]};
    }
    if (K::act1(e)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { v0[j] = gelu_tanh_fast(v0[j]); v1[j] = gelu_tanh_fast(v1[j]); }
    } else if (K::act2(e)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { v0[j] *= gelu_tanh_grad_fast(o.x0[j]); v1[j] *= gelu_tanh_grad_fast(o.x1[j]); }
    } else if (K::gate(e)) {
        v0 *= o.x0;
        v1 *= o.x1;
    }
    if (K::resid(e) || K::rowadd(e)) { v0 += o.y0; v1 += o.y1; }
    if (K::out_f32(e)) {
        if (K::beta(e)) {      // c_f32 points at a valid row group (mirrored beyond the edge)
            v0 += e.beta * load4(c_f32);
            v1 += e.beta * load4(c_f32 + 4);
        }
        const unsigned bo = ok ? 4u * (unsigned)loc : EPI_OOB;
        buf_store16(rs_c, bo, v0, !e.nt_off);
        buf_store16(rs_c, ok ? bo + 16u : EPI_OOB, v1, !e.nt_off);
    } else {
        bf16x8 r = {(bf16_t)v0[0], (bf16_t)v0[1], (bf16_t)v0[2], (bf16_t)v0[3], (bf16_t)v1[0], (bf16_t)v1[1], (bf16_t)v1[2], (bf16_t)v1[3]};
        if (K::out_q) {
            // exactly what vaw_fp8_quantize_delayed makes of the bf16 tensor this launch would otherwise have written
            v0 = f32x4{(float)r[0], (float)r[1], (float)r[2], (float)r[3]};
            v1 = f32x4{(float)r[4], (float)r[5], (float)r[6], (float)r[7]};
            if (ok) {
#pragma unroll
                for (int j = 0; j < 4; ++j) qmax = fmaxf(qmax, fmaxf(fabsf(v0[j]), fabsf(v1[j])));
            }
            const float fm = e.q_e5m2 ? 57344.f : 448.f;
            f32x4 q0 = v0 * e.q_inv, q1 = v1 * e.q_inv;
#pragma unroll
            for (int j = 0; j < 4; ++j) { q0[j] = __builtin_amdgcn_fmed3f(q0[j], -fm, fm); q1[j] = __builtin_amdgcn_fmed3f(q1[j], -fm, fm); }
            unsigned w0 = 0, w1 = 0;
            if (e.q_e5m2) {
                w0 = __builtin_amdgcn_cvt_pk_bf8_f32(q0[0], q0[1], w0, false); w0 = __builtin_amdgcn_cvt_pk_bf8_f32(q0[2], q0[3], w0, true);
                w1 = __builtin_amdgcn_cvt_pk_bf8_f32(q1[0], q1[1], w1, false); w1 = __builtin_amdgcn_cvt_pk_bf8_f32(q1[2], q1[3], w1, true);
            } else {
                w0 = __builtin_amdgcn_cvt_pk_fp8_f32(q0[0], q0[1], w0, false); w0 = __builtin_amdgcn_cvt_pk_fp8_f32(q0[2], q0[3], w0, true);
                w1 = __builtin_amdgcn_cvt_pk_fp8_f32(q1[0], q1[1], w1, false); w1 = __builtin_amdgcn_cvt_pk_fp8_f32(q1[2], q1[3], w1, true);
            }
            typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
            __builtin_amdgcn_raw_buffer_store_b64(u32x2_t{w0, w1}, rs_c, ok ? (unsigned)loc : EPI_OOB, 0, 0);
        } else {
            buf_store16(rs_c, ok ? 2u * (unsigned)loc : EPI_OOB, r, !e.nt_off);
            if (K::may_colsum) {
                v0 = f32x4{(float)r[0], (float)r[1], (float)r[2], (float)r[3]};   // what a later reader of C sees
                v1 = f32x4{(float)r[4], (float)r[5], (float)r[6], (float)r[7]};
            }
        }
    }
}
