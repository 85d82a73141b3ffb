// Row-statistics kernels of the DiT block: LayerNorm+adaLN-modulate forward/backward, gated-residual
// backward, and column sums (bias gradients).  All HBM-bound: one wave owns one token row, the row lives
// in registers between the statistics passes (one read of x, one write of the output), reductions are
// wavefront shuffles, and per-sample sums over tokens are combined through LDS in a fixed wave order
// (bitwise reproducible; no float atomics).
#include <stdio.h>
#include <stdlib.h>

#include "common.h"

// NV = number of 256-column slabs a lane walks (D <= 256*NV), D % 4 == 0.
template <typename T, int NV>
__global__ void __launch_bounds__(256)
ln_modulate_fwd_kernel(const float* __restrict__ x, const float* __restrict__ shift, const float* __restrict__ scale,
                       int64_t mod_ld, T* __restrict__ out, float* __restrict__ mean_out, float* __restrict__ rstd_out,
                       int64_t M, int Tt, int D, float eps, unsigned char* __restrict__ q_out = nullptr,
                       float* __restrict__ q_state = nullptr, int q_e5m2 = 0) {
    const int lane = threadIdx.x & 63;
    // one wave per token row; a wave walks rows blockIdx.x * 4 + wave, + 4 * gridDim.x, ... (the bf16 / f32 launches give every
    // row its own wave; the fp8 launch caps the grid so that each wave folds MANY rows' max |x| into the tensor's running max
    // with one atomic at its end: 32768 waves polling one address cost more than the whole pass -- 104 vs 46 us on DiT-XL/2)
    const float q_inv = q_out ? 1.f / q_state[0] : 1.f;
    float am = 0.f;
    // every launch caps its grid (8 workgroups per CU): a wave walks several rows and requests the NEXT row before it works on the
    // current one -- a row is one 16-byte load per lane and slab, then two wave reductions: with one row per wave (round 3) the
    // only thing in flight during the reductions was other waves' rows, and 4096 three-microsecond workgroups queued behind the
    // dispatcher (3.9-4.2 TB/s of the 6 B/elem).  Row bases are wave-uniform, lanes beyond D re-read column 0 (masked below).
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t row0 = (int64_t)blockIdx.x * 4 + wave, rstep = (int64_t)gridDim.x * 4;
    unsigned colc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const unsigned c = (unsigned)(i * 64 + lane) * 4u;
        colc[i] = c < (unsigned)D ? c : 0u;
    }
    f32x4 nx[NV];
    if (row0 < M) {
#pragma unroll
        for (int i = 0; i < NV; ++i) nx[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(x + row0 * D + colc[i]));
    }
    for (int64_t row = row0; row < M; row += rstep) {
        const int b = (int)(row / Tt);
        f32x4 v[NV];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            v[i] = c < D ? nx[i] : f32x4{0, 0, 0, 0};
            s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
        }
        if (row + rstep < M) {
#pragma unroll
            for (int i = 0; i < NV; ++i) nx[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(x + (row + rstep) * D + colc[i]));
        }
        const float mean = wave_sum(s) / (float)D;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            if (c < D) {
                f32x4 d = v[i] - mean;
                q += d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3];
            }
        }
        const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
        if (lane == 0) {
            mean_out[row] = mean;
            rstd_out[row] = rstd;
        }
        const float* sh = shift + (int64_t)b * mod_ld;
        const float* sc = scale + (int64_t)b * mod_ld;
        if (q_out) {      // fp8 mode: the row goes out as fp8 bytes of its bf16 rounding (the bf16 tensor has no other reader)
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = (i * 64 + lane) * 4;
                if (c < D) {
                    f32x4 xh = (v[i] - mean) * rstd;
                    *reinterpret_cast<unsigned*>(q_out + row * D + c) = fp8_word_of_bf16(xh * (1.f + load4(sc + c)) + load4(sh + c), q_inv, q_e5m2, am);
                }
            }
            continue;
        }
        T* orow = out + row * D;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            if (c < D) {
                f32x4 xh = (v[i] - mean) * rstd;
                store4(orow + c, xh * (1.f + load4(sc + c)) + load4(sh + c));
            }
        }
    }
    if (q_out) {          // one look at the running max per WORKGROUP (the four waves' maxima meet in LDS)
        __shared__ float wmax[4];
        am = wave_max(am);
        if (lane == 0) wmax[threadIdx.x >> 6] = am;
        __syncthreads();
        if (threadIdx.x < 64) fp8_amax_commit(fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3])), q_state + 1, lane);
    }
}

// One block per sample; NW waves stride over that sample's T rows.
// FUSE (LayerNorm mode only): the gate backward of the branch that PRECEDES this LayerNorm in the forward pass (its gradient is
// the dres row this kernel has just produced) runs on the row while it is still in registers -- dy = dres * gate, dgate += dres * y,
// column sums of dy -- instead of re-reading dres in a launch of its own (bitwise the same results: same values, same order).
template <typename T, int NV, bool GATE_ONLY, bool QOUT = false, bool FUSE = false>
__global__ void __launch_bounds__((FUSE && NV > 3) ? 512 : 1024)
row_bwd_kernel(const T* __restrict__ dout, const float* __restrict__ x, const float* __restrict__ mean,
               const float* __restrict__ rstd, const float* __restrict__ scale, int64_t mod_ld,
               const float* __restrict__ dres_in, float* __restrict__ dx, float* __restrict__ dshift,
               float* __restrict__ dscale, int64_t dmod_ld, int Tt, int D,
               // GATE_ONLY operands
               const float* __restrict__ dres, const T* __restrict__ y, const float* __restrict__ gate,
               T* __restrict__ dy, float* __restrict__ dgate, float* __restrict__ dy_colpart,
               // gridDim.y > 1: the sample's rows are cut into chunks of rows_per_chunk, one workgroup each; the per-sample
               // column sums then go to part[chunk][b][2][D] and row_bwd_finish_kernel folds the chunks in order
               int rows_per_chunk, float* __restrict__ part,
               // GATE_ONLY, fp8 mode: dy goes out as fp8 bytes of its bf16 rounding instead (q_state: its delayed-scaling state)
               unsigned char* __restrict__ q_out = nullptr, float* __restrict__ q_state = nullptr, int q_e5m2 = 0) {
    static_assert(!(FUSE && GATE_ONLY), "FUSE extends the LayerNorm mode");
    constexpr int NQ = FUSE ? 4 : 2;                               // per-sample column sums carried
    extern __shared__ __attribute__((aligned(16))) float lds[];   // [NQ][D]
    const float q_inv = ((GATE_ONLY || FUSE) && QOUT) ? 1.f / q_state[0] : 1.f;
    float q_am = 0.f;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int b = blockIdx.x;
    const int t_begin = blockIdx.y * rows_per_chunk;
    const int t_end = t_begin + rows_per_chunk < Tt ? t_begin + rows_per_chunk : Tt;
    f32x4 acc0[NV], acc1[NV], sc[NV];
    f32x4 acc2[FUSE ? NV : 1], acc3[FUSE ? NV : 1];
    // FUSE: four accumulator sets fill the 128 registers a 1024-thread workgroup may use; the sample's (1 + scale) and gate rows
    // then stay in LDS ([NQ][D] sums first, then these two rows) and are re-read per token row
    float* const sc_lds = lds + NQ * D;
    float* const gsc_lds = sc_lds + D;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        acc0[i] = f32x4{0, 0, 0, 0};
        acc1[i] = f32x4{0, 0, 0, 0};
        const int c = (i * 64 + lane) * 4;
        const float* src = GATE_ONLY ? gate : scale;
        sc[i] = c < D ? load4(src + (int64_t)b * mod_ld + c) : f32x4{0, 0, 0, 0};
        if (FUSE) {
            acc2[i] = acc3[i] = f32x4{0, 0, 0, 0};
            if (wid == 0 && c < D) {
                store4(sc_lds + c, 1.f + sc[i]);
                store4(gsc_lds + c, load4(gate + (int64_t)b * mod_ld + c));
            }
        }
    }
    if (FUSE) __syncthreads();
    for (int t = t_begin + wid; t < t_end; t += nw) {
        const int64_t row = (int64_t)b * Tt + t;
        if (GATE_ONLY) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = (i * 64 + lane) * 4;
                if (c < D) {
                    f32x4 g = load4(dres + row * D + c);
                    f32x4 yv = load4(y + row * D + c);
                    f32x4 d = g * sc[i];
                    if (QOUT) *reinterpret_cast<unsigned*>(q_out + row * D + c) = fp8_word_of_bf16(d, q_inv, q_e5m2, q_am);
                    else store4(dy + row * D + c, d);
                    acc0[i] += g * yv;
                    f32x4 dr = {to_f32(from_f32<T>(d[0])), to_f32(from_f32<T>(d[1])), to_f32(from_f32<T>(d[2])), to_f32(from_f32<T>(d[3]))};
                    acc1[i] += dr;                        // sum the values as stored (bf16-rounded in throughput mode)
                }
            }
        } else {
            const float mu = mean[row], rs = rstd[row];
            f32x4 g[NV], xh[NV];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = (i * 64 + lane) * 4;
                if (c < D) {
                    f32x4 d = load4(dout + row * D + c);
                    xh[i] = (load4(x + row * D + c) - mu) * rs;
                    acc0[i] += d;            // dshift
                    acc1[i] += d * xh[i];    // dscale
                    g[i] = d * (FUSE ? load4(sc_lds + c) : 1.f + sc[i]);
                    s1 += g[i][0] + g[i][1] + g[i][2] + g[i][3];
                    f32x4 gx = g[i] * xh[i];
                    s2 += gx[0] + gx[1] + gx[2] + gx[3];
                } else {
                    g[i] = xh[i] = f32x4{0, 0, 0, 0};
                }
            }
            const float c1 = wave_sum(s1) / (float)D, c2 = wave_sum(s2) / (float)D;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = (i * 64 + lane) * 4;
                if (c < D) {
                    f32x4 r = (g[i] - c1 - xh[i] * c2) * rs;
                    if (dres_in) r += load4(dres_in + row * D + c);
                    store4(dx + row * D + c, r);
                    if (FUSE) {      // the gate backward of vaw_gate_bwd, on the row just produced
                        const f32x4 yv = load4(y + row * D + c);
                        const f32x4 d = r * load4(gsc_lds + c);
                        if (QOUT) *reinterpret_cast<unsigned*>(q_out + row * D + c) = fp8_word_of_bf16(d, q_inv, q_e5m2, q_am);
                        else store4(dy + row * D + c, d);
                        acc2[i] += r * yv;
                        f32x4 dr = {to_f32(from_f32<T>(d[0])), to_f32(from_f32<T>(d[1])), to_f32(from_f32<T>(d[2])), to_f32(from_f32<T>(d[3]))};
                        acc3[i] += dr;
                    }
                }
            }
        }
    }
    if ((GATE_ONLY || FUSE) && QOUT) fp8_amax_commit(q_am, q_state + 1, lane);
    // fixed-order combine of the per-wave column sums
    float* s0 = lds;
    float* s1p = lds + D;
    for (int i = threadIdx.x; i < NQ * D; i += blockDim.x) lds[i] = 0.f;
    __syncthreads();
    for (int w = 0; w < nw; ++w) {
        if (wid == w) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = (i * 64 + lane) * 4;
                if (c < D) {
                    store4(s0 + c, load4(s0 + c) + acc0[i]);
                    store4(s1p + c, load4(s1p + c) + acc1[i]);
                    if (FUSE) {
                        store4(lds + 2 * D + c, load4(lds + 2 * D + c) + acc2[i]);
                        store4(lds + 3 * D + c, load4(lds + 3 * D + c) + acc3[i]);
                    }
                }
            }
        }
        __syncthreads();
    }
    if (part) {
        float* dst = part + ((int64_t)blockIdx.y * gridDim.x + b) * NQ * D;
        for (int c = threadIdx.x; c < NQ * D; c += blockDim.x) dst[c] = lds[c];
        return;
    }
    for (int c = threadIdx.x; c < D; c += blockDim.x) {
        if (GATE_ONLY) {
            dgate[(int64_t)b * dmod_ld + c] = s0[c];
            if (dy_colpart) dy_colpart[(int64_t)b * D + c] = s1p[c];
        } else {
            dshift[(int64_t)b * dmod_ld + c] = s0[c];
            dscale[(int64_t)b * dmod_ld + c] = s1p[c];
            if (FUSE) {
                dgate[(int64_t)b * dmod_ld + c] = lds[2 * D + c];
                if (dy_colpart) dy_colpart[(int64_t)b * D + c] = lds[3 * D + c];
            }
        }
    }
}

// ---- the fused LayerNorm-backward + gate-backward pass for bf16 rows up to 1280 wide (DiT-S / B / L / XL), rebuilt around memory-level
// parallelism (round 4).  row_bwd_kernel<FUSE> above keeps four accumulator sets in registers: at 1024 threads that leaves a wave
// no room to have more than one slab of one row in flight, and the compiler ends every `if (c < D)` block that holds loads with a
// wait for them -- 3.8-4.2 TB/s of its 18 B per element, 89 % of the wave time parked on memory (PMC, round 3).  Here:
//   * 8 waves per workgroup and up to 256 registers each; the four per-sample column sums live in a per-wave LDS slab [4][D]
//     (same additions in the same order as the register version, folded over the waves in wave order at the end);
//   * all four operand rows of a token row (144 bytes per lane) are requested at once, non-temporally, with wave-uniform row
//     bases and one unsigned per-lane column offset (lanes beyond D re-read column 0 and are masked out: no branch around a load);
//   * the NEXT row of the wave is requested before the current one is worked on (two register sets, loop unrolled by two), so a
//     wave always has 144-288 bytes per lane in flight and the two reductions of a row no longer expose a memory latency.
// Arithmetic and its order are those of vaw_ln_modulate_bwd followed by vaw_gate_bwd at 8 waves per workgroup (bitwise: tests).
template <int NV, bool QOUT>
__global__ void __launch_bounds__(512)
row_bwd_fuse8_kernel(const bf16_t* __restrict__ dout, const float* __restrict__ x, const float* __restrict__ mean,
                     const float* __restrict__ rstd, const float* __restrict__ scale, int64_t mod_ld,
                     const float* __restrict__ dres_in, float* __restrict__ dx, float* __restrict__ dshift,
                     float* __restrict__ dscale, int64_t dmod_ld, int Tt, int D, const bf16_t* __restrict__ y,
                     const float* __restrict__ gate, bf16_t* __restrict__ dy, float* __restrict__ dgate, float* __restrict__ dy_colpart,
                     int rows_per_chunk, float* __restrict__ part, unsigned char* __restrict__ q_out, float* __restrict__ q_state,
                     int q_e5m2) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // (1 + scale) row | gate row | [waves][4][D] sums
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    const int b = blockIdx.x;
    const int t_begin = blockIdx.y * rows_per_chunk;
    const int t_end = t_begin + rows_per_chunk < Tt ? t_begin + rows_per_chunk : Tt;
    float* const sc_lds = lds;
    float* const gsc_lds = lds + D;
    float* const slab = lds + 2 * D + wid * 4 * D;
    const float q_inv = QOUT ? 1.f / q_state[0] : 1.f;
    float q_am = 0.f;
    bool act[NV];
    unsigned col[NV], colc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        col[i] = (unsigned)(i * 64 + lane) * 4u;
        act[i] = col[i] < (unsigned)D;
        colc[i] = act[i] ? col[i] : 0u;
        if (act[i]) {
#pragma unroll
            for (int q = 0; q < 4; ++q) store4(slab + q * D + col[i], f32x4{0, 0, 0, 0});
            if (wid == 0) {
                store4(sc_lds + col[i], 1.f + load4(scale + (int64_t)b * mod_ld + col[i]));
                store4(gsc_lds + col[i], load4(gate + (int64_t)b * mod_ld + col[i]));
            }
        }
    }
    __syncthreads();
    struct RowRegs {
        bf16x4 dq[NV], yq[NV];
        f32x4 xv[NV], rv[NV];
        float mu, rs;
    };
    auto issue = [&](RowRegs& R, int t) __attribute__((always_inline)) {
        const int64_t row = (int64_t)b * Tt + t;
        const bf16_t* const dout_r = dout + row * D;
        const float* const x_r = x + row * D;
        const float* const res_r = dres_in ? dres_in + row * D : x_r;
        const bf16_t* const y_r = y + row * D;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            R.dq[i] = __builtin_nontemporal_load(reinterpret_cast<const bf16x4*>(dout_r + colc[i]));
            R.xv[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(x_r + colc[i]));
            R.rv[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(res_r + colc[i]));
            R.yq[i] = __builtin_nontemporal_load(reinterpret_cast<const bf16x4*>(y_r + colc[i]));
        }
        R.mu = mean[row];
        R.rs = rstd[row];
    };
    auto process = [&](RowRegs& R, int t) __attribute__((always_inline)) {
        const int64_t row = (int64_t)b * Tt + t;
        const float mu = R.mu, rs = R.rs;
        const f32x4 z = {0, 0, 0, 0};
        f32x4 xh[NV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const f32x4 d = act[i] ? f32x4{(float)R.dq[i][0], (float)R.dq[i][1], (float)R.dq[i][2], (float)R.dq[i][3]} : z;
            xh[i] = act[i] ? (R.xv[i] - mu) * rs : z;
            const f32x4 g = d * load4(sc_lds + colc[i]);
            s1 += g[0] + g[1] + g[2] + g[3];
            const f32x4 gx = g * xh[i];
            s2 += gx[0] + gx[1] + gx[2] + gx[3];
            if (act[i]) {
                float* p0 = slab + col[i];
                store4(p0, load4(p0) + d);                    // dshift
                store4(p0 + D, load4(p0 + D) + d * xh[i]);    // dscale
            }
        }
        const float c1 = wave_sum(s1) / (float)D, c2 = wave_sum(s2) / (float)D;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if (!act[i]) continue;
            const f32x4 d0 = {(float)R.dq[i][0], (float)R.dq[i][1], (float)R.dq[i][2], (float)R.dq[i][3]};
            const f32x4 g = d0 * load4(sc_lds + col[i]);
            f32x4 r = (g - c1 - xh[i] * c2) * rs;
            if (dres_in) r += R.rv[i];
            __builtin_nontemporal_store(r, reinterpret_cast<f32x4*>(dx + row * D + col[i]));
            const f32x4 yv = {(float)R.yq[i][0], (float)R.yq[i][1], (float)R.yq[i][2], (float)R.yq[i][3]};
            const f32x4 d = r * load4(gsc_lds + col[i]);
            if (QOUT) *reinterpret_cast<unsigned*>(q_out + row * D + col[i]) = fp8_word_of_bf16(d, q_inv, q_e5m2, q_am);
            else store4(dy + row * D + col[i], d);
            const f32x4 dr = {(float)(bf16_t)d[0], (float)(bf16_t)d[1], (float)(bf16_t)d[2], (float)(bf16_t)d[3]};
            float* p2 = slab + 2 * D + col[i];
            store4(p2, load4(p2) + r * yv);                   // dgate
            store4(p2 + D, load4(p2 + D) + dr);               // column sums of dy as stored
        }
    };
    {
        RowRegs A, Bq;
        int t = t_begin + wid;
        if (t < t_end) issue(A, t);
        while (t < t_end) {
            const int t1 = t + nw;
            if (t1 < t_end) issue(Bq, t1);
            process(A, t);
            if (t1 >= t_end) break;
            const int t2 = t1 + nw;
            if (t2 < t_end) issue(A, t2);
            process(Bq, t1);
            t = t2;
        }
    }
    if (QOUT) fp8_amax_commit(q_am, q_state + 1, lane);
    __syncthreads();
    // fold the waves' slabs in wave order (the order of row_bwd_kernel's turn-taking combine)
    for (int idx = threadIdx.x; idx < 4 * D; idx += blockDim.x) {
        const int q = idx / D, c = idx - q * D;
        float s = 0.f;
        for (int w = 0; w < nw; ++w) s += lds[2 * D + (w * 4 + q) * D + c];
        if (part) part[((int64_t)blockIdx.y * gridDim.x + b) * 4 * D + idx] = s;
        else if (q == 0) dshift[(int64_t)b * dmod_ld + c] = s;
        else if (q == 1) dscale[(int64_t)b * dmod_ld + c] = s;
        else if (q == 2) dgate[(int64_t)b * dmod_ld + c] = s;
        else if (dy_colpart) dy_colpart[(int64_t)b * D + c] = s;
    }
}
template <int NV, bool QOUT>
static size_t row_fuse8_lds(int D, int block) {
    const size_t lds = (2 + 4 * (size_t)(block / 64)) * (size_t)D * sizeof(float);
    static bool done = false;
    if (!done && lds > 64 * 1024) {
        const hipError_t rc = hipFuncSetAttribute((const void*)row_bwd_fuse8_kernel<NV, QOUT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (rc != hipSuccess) fprintf(stderr, "row_fuse8_lds: hipFuncSetAttribute -> %s\n", hipGetErrorString(rc));
        done = true;
    }
    return lds;
}

// out0[b][c] (row stride ld0) = sum_chunk part[chunk][b][0][c], out1 likewise from [1] (row stride ld1; may be NULL); the fused
// kernel's part rows carry four quantities: out2 / out3 from [2] / [3]
__global__ void row_bwd_finish_kernel(const float* __restrict__ part, int NC, int B, int D, float* __restrict__ out0, int64_t ld0,
                                      float* __restrict__ out1, int64_t ld1, int NQ = 2, float* __restrict__ out2 = nullptr,
                                      int64_t ld2 = 0, float* __restrict__ out3 = nullptr, int64_t ld3 = 0) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * D) return;
    const int b = (int)(i / D), c = (int)(i % D);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int ch = 0; ch < NC; ++ch) {
        const float* src = part + ((int64_t)ch * B + b) * NQ * D;
        a0 += src[c];
        a1 += src[D + c];
        if (NQ == 4) {
            a2 += src[2 * D + c];
            a3 += src[3 * D + c];
        }
    }
    out0[(int64_t)b * ld0 + c] = a0;
    if (out1) out1[(int64_t)b * ld1 + c] = a1;
    if (out2) out2[(int64_t)b * ld2 + c] = a2;
    if (out3) out3[(int64_t)b * ld3 + c] = a3;
}

static int pick_nv(int D) { return (D + 255) / 256; }
// chunks per sample so that small batches still fill the chip: aim at >= 256 workgroups (one per CU; at B >= 256 the
// single 1024-thread workgroup per sample already runs at 4.5 TB/s and the extra fold launch only costs), >= 8 rows per chunk
static int pick_chunks(int B, int Tt, bool have_ws, int target_wgs = 256) {
    if (!have_ws) return 1;
    int nc = (target_wgs + B - 1) / B;
    const int max_nc = Tt / 8 > 0 ? Tt / 8 : 1;
    if (nc > max_nc) nc = max_nc;
    return nc < 1 ? 1 : nc;
}
extern "C" int64_t vaw_row_bwd_workspace_floats(int B, int T, int D) { return (int64_t)pick_chunks(B, T, true, 512) * B * 4 * D; }

// (rows up to 1280 wide: 8 waves everywhere, so that the fused pass -- row_bwd_fuse8_kernel -- and the pair of kernels it replaces cut a
// sample's rows into the same per-wave partial sums and stay bitwise equal)
static int row_waves(int nv, int wide_default) { return nv <= 5 ? 8 : wide_default; }
static int pick_block(int Tt, int max_waves = 16) {
    int nw = Tt < max_waves ? Tt : max_waves;
    if (nw < 1) nw = 1;
    return nw * 64;
}

#define DISPATCH_NV(nv, CALL)                                       \
    switch (nv) {                                                   \
        case 1: { constexpr int NV = 1; CALL; } break;              \
        case 2: { constexpr int NV = 2; CALL; } break;              \
        case 3: { constexpr int NV = 3; CALL; } break;              \
        case 4: { constexpr int NV = 4; CALL; } break;              \
        case 5: { constexpr int NV = 5; CALL; } break;              \
        case 6: { constexpr int NV = 6; CALL; } break;              \
        default: { constexpr int NV = 8; CALL; } break;             \
    }

extern "C" int vaw_ln_modulate_fwd(vaw_dtype dt, const float* x, const float* shift, const float* scale, int64_t mod_ld,
                                   void* out, float* mean, float* rstd, int B, int T, int D, float eps,
                                   vaw_stream stream) {
    VAW_CHECK_ARG(B > 0 && T > 0 && D > 0 && D % 4 == 0 && D <= 2048 && mod_ld % 4 == 0,
                  "ln_modulate_fwd: need D%%4==0, D<=2048, mod_ld%%4==0 (D=%d)", D);
    const int64_t M = (int64_t)B * T;
    // one resident round of workgroups (what the registers allow per CU x 256 CUs: 7 x 256 for 768-wide bf16 rows -- a grid of
    // 8 per CU left the eighth to a second, nearly empty round), every wave several rows: see the kernel
    static int per_cu[2][9] = {};
    const int nvk = pick_nv(D) <= 6 ? pick_nv(D) : 8, ti = dt == VAW_F32 ? 0 : 1;
    if (!per_cu[ti][nvk]) {
        int nb = 0;
        if (dt == VAW_F32) { DISPATCH_NV(nvk, (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, ln_modulate_fwd_kernel<float, NV>, 256, 0)); }
        else { DISPATCH_NV(nvk, (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, ln_modulate_fwd_kernel<bf16_t, NV>, 256, 0)); }
        per_cu[ti][nvk] = nb > 0 && nb <= 8 ? nb : 8;
    }
    const int64_t cap = 256LL * per_cu[ti][nvk];
    const int grid = (int)(ceil_div(M, 4) < cap ? ceil_div(M, 4) : cap);
    hipStream_t s = (hipStream_t)stream;
    if (dt == VAW_F32) {
        DISPATCH_NV(pick_nv(D), (ln_modulate_fwd_kernel<float, NV><<<grid, 256, 0, s>>>(x, shift, scale, mod_ld, (float*)out, mean, rstd, M, T, D, eps)));
    } else {
        DISPATCH_NV(pick_nv(D), (ln_modulate_fwd_kernel<bf16_t, NV><<<grid, 256, 0, s>>>(x, shift, scale, mod_ld, (bf16_t*)out, mean, rstd, M, T, D, eps)));
    }
    VAW_CHECK_LAUNCH("ln_modulate_fwd");
    return VAW_OK;
}

// fp8 mode: the same forward with the output as fp8 bytes [B*T][D] of the bf16 roundings (delayed scaling: scale in q_state[0],
// running max |x| folded into q_state[1]); bf16 arithmetic path only.
extern "C" int vaw_ln_modulate_fwd_fp8(const float* x, const float* shift, const float* scale, int64_t mod_ld, void* q_out,
                                       float* q_state, vaw_dtype q_format, float* mean, float* rstd, int B, int T, int D, float eps,
                                       vaw_stream stream) {
    VAW_CHECK_ARG(B > 0 && T > 0 && D > 0 && D % 4 == 0 && D <= 2048 && mod_ld % 4 == 0 && q_out && q_state,
                  "ln_modulate_fwd_fp8: need D%%4==0, D<=2048, mod_ld%%4==0 (D=%d)", D);
    VAW_CHECK_ARG(q_format == VAW_FP8 || q_format == VAW_BF8, "ln_modulate_fwd_fp8: q_format");
    const int64_t M = (int64_t)B * T;
    const int grid = ceil_div(M, 4) < 2048 ? ceil_div(M, 4) : 2048;       // 8 workgroups per CU, every wave several rows: see the kernel
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_NV(pick_nv(D), (ln_modulate_fwd_kernel<bf16_t, NV><<<grid, 256, 0, s>>>(x, shift, scale, mod_ld, nullptr, mean, rstd, M, T, D, eps,
                                                                                  (unsigned char*)q_out, q_state, q_format == VAW_BF8)));
    VAW_CHECK_LAUNCH("ln_modulate_fwd_fp8");
    return VAW_OK;
}

extern "C" int vaw_ln_modulate_bwd(vaw_dtype dt, const void* dout, const float* x, const float* mean, const float* rstd,
                                   const float* scale, int64_t mod_ld, const float* dres_in, float* dx, float* dshift,
                                   float* dscale, int64_t dmod_ld, int B, int T, int D, float* workspace,
                                   int64_t workspace_floats, vaw_stream stream) {
    VAW_CHECK_ARG(B > 0 && T > 0 && D > 0 && D % 4 == 0 && D <= 2048 && mod_ld % 4 == 0,
                  "ln_modulate_bwd: need D%%4==0, D<=2048, mod_ld%%4==0 (D=%d)", D);
    hipStream_t s = (hipStream_t)stream;
    int nc = pick_chunks(B, T, workspace != nullptr);
    if (nc > 1 && workspace_floats < (int64_t)nc * B * 2 * D) nc = 1;
    const int rpc = (T + nc - 1) / nc;
    float* part = nc > 1 ? workspace : nullptr;
    const int block = pick_block(rpc, row_waves(pick_nv(D), 16));
    const size_t lds = 2 * (size_t)D * sizeof(float);
    dim3 grid(B, nc);
    if (dt == VAW_F32) {
        DISPATCH_NV(pick_nv(D), (row_bwd_kernel<float, NV, false><<<grid, block, lds, s>>>((const float*)dout, x, mean, rstd, scale, mod_ld, dres_in, dx, dshift, dscale, dmod_ld, T, D, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, rpc, part)));
    } else {
        DISPATCH_NV(pick_nv(D), (row_bwd_kernel<bf16_t, NV, false><<<grid, block, lds, s>>>((const bf16_t*)dout, x, mean, rstd, scale, mod_ld, dres_in, dx, dshift, dscale, dmod_ld, T, D, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, rpc, part)));
    }
    if (nc > 1) row_bwd_finish_kernel<<<ceil_div((int64_t)B * D, 256), 256, 0, s>>>(part, nc, B, D, dshift, dmod_ld, dscale, dmod_ld);
    VAW_CHECK_LAUNCH("ln_modulate_bwd");
    return VAW_OK;
}

// vaw_ln_modulate_bwd followed by vaw_gate_bwd of the branch in front of this LayerNorm, as ONE pass (row_bwd_kernel<FUSE>):
// dx = the residual-stream gradient as before; dy_next = dx * gate_next (act dtype), dgate_next[b] = sum_t dx * y_next,
// dy_colsum_partial [B][D] = per-sample sum_t dy_next (as stored).  Bitwise equal to the two separate launches.
extern "C" int vaw_ln_modulate_bwd_gate(vaw_dtype dt, const void* dout, const float* x, const float* mean, const float* rstd,
                                        const float* scale, int64_t mod_ld, const float* dres_in, float* dx, float* dshift,
                                        float* dscale, int64_t dmod_ld, const void* y_next, const float* gate_next, void* dy_next,
                                        float* dgate_next, float* dy_colsum_partial, int B, int T, int D, float* workspace,
                                        int64_t workspace_floats, vaw_stream stream) {
    VAW_CHECK_ARG(B > 0 && T > 0 && D > 0 && D % 4 == 0 && D <= 2048 && mod_ld % 4 == 0,
                  "ln_modulate_bwd_gate: need D%%4==0, D<=2048, mod_ld%%4==0 (D=%d)", D);
    VAW_CHECK_ARG(y_next && gate_next && dy_next && dgate_next, "ln_modulate_bwd_gate: the gate operands are required");
    hipStream_t s = (hipStream_t)stream;
    const int nv = pick_nv(D);
    // wide rows (D > 768: four accumulator sets need 176-190 registers) run 512-thread workgroups, two per CU: cut the samples
    // into enough chunks for 512 of them (DiT-XL/2 at 128 images: 200 us with 256 workgroups at 3.2 TB/s)
    int nc = pick_chunks(B, T, workspace != nullptr, nv > 3 ? 512 : 256);
    if (nc > 1 && workspace_floats < (int64_t)nc * B * 4 * D) nc = 1;
    const int rpc = (T + nc - 1) / nc;
    float* part = nc > 1 ? workspace : nullptr;
    const int block = pick_block(rpc, 8);
    const size_t lds = 6 * (size_t)D * sizeof(float);
    dim3 grid(B, nc);
    if (dt == VAW_F32) {
        DISPATCH_NV(nv, (row_bwd_kernel<float, NV, false, false, true><<<grid, block, lds, s>>>((const float*)dout, x, mean, rstd, scale, mod_ld, dres_in, dx, dshift, dscale, dmod_ld, T, D, nullptr, (const float*)y_next, gate_next, (float*)dy_next, dgate_next, dy_colsum_partial, rpc, part)));
    } else if (nv <= 5) {
#define LAUNCH_FUSE8(NVv)                                                                                                              \
    row_bwd_fuse8_kernel<NVv, false><<<grid, block, row_fuse8_lds<NVv, false>(D, block), s>>>(                                          \
        (const bf16_t*)dout, x, mean, rstd, scale, mod_ld, dres_in, dx, dshift, dscale, dmod_ld, T, D, (const bf16_t*)y_next, gate_next, \
        (bf16_t*)dy_next, dgate_next, dy_colsum_partial, rpc, part, nullptr, nullptr, 0)
        if (nv == 1) LAUNCH_FUSE8(1); else if (nv == 2) LAUNCH_FUSE8(2); else if (nv == 3) LAUNCH_FUSE8(3); else if (nv == 4) LAUNCH_FUSE8(4); else LAUNCH_FUSE8(5);
    } else {
        DISPATCH_NV(nv, (row_bwd_kernel<bf16_t, NV, false, false, true><<<grid, block, lds, s>>>((const bf16_t*)dout, x, mean, rstd, scale, mod_ld, dres_in, dx, dshift, dscale, dmod_ld, T, D, nullptr, (const bf16_t*)y_next, gate_next, (bf16_t*)dy_next, dgate_next, dy_colsum_partial, rpc, part)));
    }
    if (nc > 1)
        row_bwd_finish_kernel<<<ceil_div((int64_t)B * D, 256), 256, 0, s>>>(part, nc, B, D, dshift, dmod_ld, dscale, dmod_ld, 4, dgate_next, dmod_ld,
                                                                             dy_colsum_partial, D);
    VAW_CHECK_LAUNCH("ln_modulate_bwd_gate");
    return VAW_OK;
}

// fp8 mode: the same fused pass with dy_next as fp8 bytes [B*T][D] of its bf16 roundings (as vaw_gate_bwd_fp8); bf16 arithmetic path.
extern "C" int vaw_ln_modulate_bwd_gate_fp8(const void* dout, const float* x, const float* mean, const float* rstd, const float* scale,
                                            int64_t mod_ld, const float* dres_in, float* dx, float* dshift, float* dscale, int64_t dmod_ld,
                                            const void* y_next, const float* gate_next, void* dy_q, float* q_state, vaw_dtype q_format,
                                            float* dgate_next, float* dy_colsum_partial, int B, int T, int D, float* workspace,
                                            int64_t workspace_floats, vaw_stream stream) {
    VAW_CHECK_ARG(B > 0 && T > 0 && D > 0 && D % 4 == 0 && D <= 2048 && mod_ld % 4 == 0,
                  "ln_modulate_bwd_gate_fp8: need D%%4==0, D<=2048, mod_ld%%4==0 (D=%d)", D);
    VAW_CHECK_ARG(y_next && gate_next && dy_q && q_state && dgate_next, "ln_modulate_bwd_gate_fp8: the gate operands are required");
    VAW_CHECK_ARG(q_format == VAW_FP8 || q_format == VAW_BF8, "ln_modulate_bwd_gate_fp8: q_format");
    hipStream_t s = (hipStream_t)stream;
    const int nv = pick_nv(D);
    // wide rows (D > 768: four accumulator sets need 176-190 registers) run 512-thread workgroups, two per CU: cut the samples
    // into enough chunks for 512 of them (DiT-XL/2 at 128 images: 200 us with 256 workgroups at 3.2 TB/s)
    int nc = pick_chunks(B, T, workspace != nullptr, nv > 3 ? 512 : 256);
    if (nc > 1 && workspace_floats < (int64_t)nc * B * 4 * D) nc = 1;
    const int rpc = (T + nc - 1) / nc;
    float* part = nc > 1 ? workspace : nullptr;
    const int block = pick_block(rpc, 8);
    const size_t lds = 6 * (size_t)D * sizeof(float);
    dim3 grid(B, nc);
    if (nv <= 5) {
#define LAUNCH_FUSE8Q(NVv)                                                                                                             \
    row_bwd_fuse8_kernel<NVv, true><<<grid, block, row_fuse8_lds<NVv, true>(D, block), s>>>(                                            \
        (const bf16_t*)dout, x, mean, rstd, scale, mod_ld, dres_in, dx, dshift, dscale, dmod_ld, T, D, (const bf16_t*)y_next, gate_next, \
        nullptr, dgate_next, dy_colsum_partial, rpc, part, (unsigned char*)dy_q, q_state, q_format == VAW_BF8)
        if (nv == 1) LAUNCH_FUSE8Q(1); else if (nv == 2) LAUNCH_FUSE8Q(2); else if (nv == 3) LAUNCH_FUSE8Q(3); else if (nv == 4) LAUNCH_FUSE8Q(4); else LAUNCH_FUSE8Q(5);
    } else {
        DISPATCH_NV(nv, (row_bwd_kernel<bf16_t, NV, false, true, true><<<grid, block, lds, s>>>((const bf16_t*)dout, x, mean, rstd, scale, mod_ld, dres_in, dx, dshift, dscale, dmod_ld, T, D, nullptr, (const bf16_t*)y_next, gate_next, nullptr, dgate_next, dy_colsum_partial, rpc, part, (unsigned char*)dy_q, q_state, q_format == VAW_BF8)));
    }
    if (nc > 1)
        row_bwd_finish_kernel<<<ceil_div((int64_t)B * D, 256), 256, 0, s>>>(part, nc, B, D, dshift, dmod_ld, dscale, dmod_ld, 4, dgate_next, dmod_ld,
                                                                             dy_colsum_partial, D);
    VAW_CHECK_LAUNCH("ln_modulate_bwd_gate_fp8");
    return VAW_OK;
}

extern "C" int vaw_gate_bwd(vaw_dtype dt, const float* dres, const void* y, const float* gate, int64_t mod_ld, void* dy,
                            float* dgate, int64_t dmod_ld, float* dy_colsum_partial, int B, int T, int D, float* workspace,
                            int64_t workspace_floats, vaw_stream stream) {
    VAW_CHECK_ARG(B > 0 && T > 0 && D > 0 && D % 4 == 0 && D <= 2048 && mod_ld % 4 == 0,
                  "gate_bwd: need D%%4==0, D<=2048, mod_ld%%4==0 (D=%d)", D);
    hipStream_t s = (hipStream_t)stream;
    int nc = pick_chunks(B, T, workspace != nullptr);
    if (nc > 1 && workspace_floats < (int64_t)nc * B * 2 * D) nc = 1;
    const int rpc = (T + nc - 1) / nc;
    float* part = nc > 1 ? workspace : nullptr;
    const int block = pick_block(rpc, row_waves(pick_nv(D), 16));
    const size_t lds = 2 * (size_t)D * sizeof(float);
    dim3 grid(B, nc);
    if (dt == VAW_F32) {
        DISPATCH_NV(pick_nv(D), (row_bwd_kernel<float, NV, true><<<grid, block, lds, s>>>(nullptr, nullptr, nullptr, nullptr, nullptr, mod_ld, nullptr, nullptr, nullptr, nullptr, dmod_ld, T, D, dres, (const float*)y, gate, (float*)dy, dgate, dy_colsum_partial, rpc, part)));
    } else {
        DISPATCH_NV(pick_nv(D), (row_bwd_kernel<bf16_t, NV, true><<<grid, block, lds, s>>>(nullptr, nullptr, nullptr, nullptr, nullptr, mod_ld, nullptr, nullptr, nullptr, nullptr, dmod_ld, T, D, dres, (const bf16_t*)y, gate, (bf16_t*)dy, dgate, dy_colsum_partial, rpc, part)));
    }
    if (nc > 1) row_bwd_finish_kernel<<<ceil_div((int64_t)B * D, 256), 256, 0, s>>>(part, nc, B, D, dgate, dmod_ld, dy_colsum_partial, D);
    VAW_CHECK_LAUNCH("gate_bwd");
    return VAW_OK;
}

// fp8 mode: gate backward with dy as fp8 bytes [B*T][D] of the bf16 roundings (as vaw_ln_modulate_fwd_fp8); y is bf16.
extern "C" int vaw_gate_bwd_fp8(const float* dres, const void* y, const float* gate, int64_t mod_ld, void* dy_q, float* q_state,
                                vaw_dtype q_format, float* dgate, int64_t dmod_ld, float* dy_colsum_partial, int B, int T, int D,
                                float* workspace, int64_t workspace_floats, vaw_stream stream) {
    VAW_CHECK_ARG(B > 0 && T > 0 && D > 0 && D % 4 == 0 && D <= 2048 && mod_ld % 4 == 0 && dy_q && q_state,
                  "gate_bwd_fp8: need D%%4==0, D<=2048, mod_ld%%4==0 (D=%d)", D);
    VAW_CHECK_ARG(q_format == VAW_FP8 || q_format == VAW_BF8, "gate_bwd_fp8: q_format");
    hipStream_t s = (hipStream_t)stream;
    int nc = pick_chunks(B, T, workspace != nullptr);
    if (nc > 1 && workspace_floats < (int64_t)nc * B * 2 * D) nc = 1;
    const int rpc = (T + nc - 1) / nc;
    float* part = nc > 1 ? workspace : nullptr;
    const int block = pick_block(rpc, row_waves(pick_nv(D), 16));
    const size_t lds = 2 * (size_t)D * sizeof(float);
    dim3 grid(B, nc);
    DISPATCH_NV(pick_nv(D), (row_bwd_kernel<bf16_t, NV, true, true><<<grid, block, lds, s>>>(nullptr, nullptr, nullptr, nullptr, nullptr, mod_ld, nullptr, nullptr, nullptr, nullptr, dmod_ld, T, D, dres, (const bf16_t*)y, gate, nullptr, dgate, dy_colsum_partial, rpc, part, (unsigned char*)dy_q, q_state, q_format == VAW_BF8)));
    if (nc > 1) row_bwd_finish_kernel<<<ceil_div((int64_t)B * D, 256), 256, 0, s>>>(part, nc, B, D, dgate, dmod_ld, dy_colsum_partial, D);
    VAW_CHECK_LAUNCH("gate_bwd_fp8");
    return VAW_OK;
}

// ---------------------------------------------------------------------------------------------
// Column sums: out[n] = beta*out[n] + sum_m X[m,n].  Stage 1: 256 columns x 512 rows per block, four row
// groups folded in LDS, one partial row per row-block written to the workspace.  Stage 2: the partial rows
// are added in a fixed order.  No float atomics: bias gradients are bitwise reproducible.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256)
colsum_partial_kernel(const T* __restrict__ X, int64_t M, int64_t N, int64_t ldx, float* __restrict__ part_out) {
    __shared__ __attribute__((aligned(16))) float part[4][256];
    const int cg = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int64_t col = (int64_t)blockIdx.x * 256 + cg * 4;
    const int64_t r0 = (int64_t)blockIdx.y * 512;
    const int64_t r1 = r0 + 512 < M ? r0 + 512 : M;
    f32x4 acc = {0, 0, 0, 0};
    if (col < N) {
        if (col + 4 <= N) {
            for (int64_t r = r0 + rg; r < r1; r += 4) acc += load4(X + r * ldx + col);
        } else {
            for (int64_t r = r0 + rg; r < r1; r += 4)
                for (int j = 0; j < 4 && col + j < N; ++j) acc[j] += to_f32(X[r * ldx + col + j]);
        }
    }
    store4(&part[rg][cg * 4], acc);
    __syncthreads();
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c < N)
        part_out[(int64_t)blockIdx.y * N + c] =
            ((part[0][threadIdx.x] + part[1][threadIdx.x]) + part[2][threadIdx.x]) + part[3][threadIdx.x];
}
// out[c] = beta*out[c] + sum_r part[r][c], r ascending within each of 32 interleaved row groups, groups folded in
// ascending order: fixed summation tree.  32 columns x 32 row groups per 1024-thread block.
__global__ void __launch_bounds__(1024)
colsum_final_kernel(const float* __restrict__ part, int64_t RB, int64_t N, float* __restrict__ out, float beta) {
    __shared__ float fold[32][33];
    const int cl = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int64_t c = (int64_t)blockIdx.x * 32 + cl;
    float acc = 0.f;
    if (c < N)
        for (int64_t r = grp; r < RB; r += 32) acc += part[r * N + c];
    fold[grp][cl] = acc;
    __syncthreads();
    if (grp == 0 && c < N) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 32; ++g) t += fold[g][cl];
        out[c] = (beta != 0.f ? beta * out[c] : 0.f) + t;
    }
}

// bf16, N % 8 == 0, 16-byte aligned rows: 128 rows x 256 columns per workgroup, every lane 16 independent 16-byte loads
// (the 512-row kernel above walks its rows with one dependent 8-byte load at a time: 1.9 TB/s on [16384][2304])
__global__ void __launch_bounds__(256)
colsum_partial_bf16x8_kernel(const bf16_t* __restrict__ X, int64_t M, int64_t N, int64_t ldx, float* __restrict__ part_out) {
    __shared__ __attribute__((aligned(16))) float part[8][256];
    const int cg = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int64_t col = (int64_t)blockIdx.x * 256 + cg * 8;
    const int64_t r0 = (int64_t)blockIdx.y * 128;
    f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
    if (col < N) {          // N % 8 == 0: a group of 8 columns is in or out as a whole
        bf16x8 v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int64_t r = r0 + rg + 8 * i;
            v[i] = r < M ? *reinterpret_cast<const bf16x8*>(X + r * ldx + col) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {        // rows ascending: fixed order
            a0 += f32x4{(float)v[i][0], (float)v[i][1], (float)v[i][2], (float)v[i][3]};
            a1 += f32x4{(float)v[i][4], (float)v[i][5], (float)v[i][6], (float)v[i][7]};
        }
    }
    store4(&part[rg][cg * 8], a0);
    store4(&part[rg][cg * 8 + 4], a1);
    __syncthreads();
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c < N) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) t += part[g][threadIdx.x];
        part_out[(int64_t)blockIdx.y * N + c] = t;
    }
}

// any alignment / any N (e.g. the 3-channel output conv): one thread per (row block, column)
template <typename T>
__global__ void colsum_partial_scalar_kernel(const T* __restrict__ X, int64_t M, int64_t N, int64_t ldx, float* __restrict__ part_out) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= N) return;
    const int64_t r0 = (int64_t)blockIdx.y * 512, r1 = r0 + 512 < M ? r0 + 512 : M;
    float acc = 0.f;
    for (int64_t r = r0; r < r1; ++r) acc += to_f32(X[r * ldx + c]);
    part_out[(int64_t)blockIdx.y * N + c] = acc;
}

// Many column-sum folds in ONE launch: job j folds R_j partial rows of N_j columns into out_j exactly as colsum_final_kernel
// would (same summation tree: bitwise the same result).  The bias gradients of all Linear layers of a group of DiT blocks are
// produced as partial rows by the kernels that make dy (gate backward, GELU' epilogue, attention backward) and folded here
// once per group instead of by one 5-microsecond launch each.
struct ReduceJobDev {
    const float* part;
    float* out;
    int64_t R, N;
    int block0, pad_;
};
__global__ void __launch_bounds__(1024)
colsum_final_batched_kernel(const ReduceJobDev* __restrict__ jobs, int n_jobs, float beta) {
    __shared__ float fold[32][33];
    int lo = 0, hi = n_jobs - 1;                      // last job with block0 <= blockIdx.x (uniform: scalar loads)
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].block0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const ReduceJobDev jb = jobs[lo];
    const int cl = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int64_t c = (int64_t)(blockIdx.x - jb.block0) * 32 + cl;
    float acc = 0.f;
    if (c < jb.N)
        for (int64_t r = grp; r < jb.R; r += 32) acc += jb.part[r * jb.N + c];
    fold[grp][cl] = acc;
    __syncthreads();
    if (grp == 0 && c < jb.N) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 32; ++g) t += fold[g][cl];
        jb.out[c] = (beta != 0.f ? beta * jb.out[c] : 0.f) + t;
    }
}

extern "C" int64_t vaw_reduce_rows_batched_desc_bytes(int n_jobs) { return (int64_t)n_jobs * (int64_t)sizeof(ReduceJobDev); }

extern "C" int vaw_reduce_rows_batched(int n_jobs, const vaw_reduce_job* jobs, float beta, void* desc_dev, int upload, vaw_stream stream) {
    VAW_CHECK_ARG(n_jobs > 0 && n_jobs <= 4096 && jobs && desc_dev, "reduce_rows_batched: bad arguments");
    static thread_local ReduceJobDev host[4096];
    int64_t blocks = 0;
    for (int j = 0; j < n_jobs; ++j) {
        VAW_CHECK_ARG(jobs[j].partial && jobs[j].out && jobs[j].R > 0 && jobs[j].N > 0, "reduce_rows_batched: job %d", j);
        host[j] = ReduceJobDev{jobs[j].partial, jobs[j].out, jobs[j].R, jobs[j].N, (int)blocks, 0};
        blocks += (jobs[j].N + 31) / 32;
        VAW_CHECK_ARG(blocks < (1 << 30), "reduce_rows_batched: too many columns");
    }
    hipStream_t s = (hipStream_t)stream;
    if (upload) {
        const hipError_t rc = vaw_upload_table(desc_dev, host, sizeof(ReduceJobDev) * n_jobs, s);
        VAW_CHECK_ARG(rc == hipSuccess, "reduce_rows_batched: descriptor upload failed: %s", hipGetErrorString(rc));
    }
    colsum_final_batched_kernel<<<(unsigned)blocks, 1024, 0, s>>>((const ReduceJobDev*)desc_dev, n_jobs, beta);
    VAW_CHECK_LAUNCH("reduce_rows_batched");
    return VAW_OK;
}

extern "C" int64_t vaw_colsum_workspace_floats(int64_t M, int64_t N) { return ((M + 127) / 128) * N; }

extern "C" int vaw_reduce_rows(const float* partial, int64_t R, int64_t N, float* out, float beta, vaw_stream stream) {
    VAW_CHECK_ARG(R > 0 && N > 0 && partial && out, "reduce_rows: bad arguments");
    colsum_final_kernel<<<ceil_div(N, 32), 1024, 0, (hipStream_t)stream>>>(partial, R, N, out, beta);
    VAW_CHECK_LAUNCH("reduce_rows");
    return VAW_OK;
}

extern "C" int vaw_colsum(vaw_dtype dt, const void* X, int64_t M, int64_t N, int64_t ldx, float* out, float beta,
                          float* workspace, int64_t workspace_floats, vaw_stream stream) {
    VAW_CHECK_ARG(M > 0 && N > 0 && ldx >= N, "colsum: bad sizes");
    const bool vec = (ldx % 4 == 0) && (((uintptr_t)X & 15) == 0);
    const bool wide = dt == VAW_BF16 && N % 8 == 0 && ldx % 8 == 0 && (((uintptr_t)X & 15) == 0) && M >= 1024;
    const int64_t RB = wide ? (M + 127) / 128 : (M + 511) / 512;
    VAW_CHECK_ARG(workspace && workspace_floats >= RB * N, "colsum: workspace too small (%ld < %ld floats)",
                  (long)workspace_floats, (long)(RB * N));
    VAW_CHECK_ARG(RB < 65536, "colsum: M too large");
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(ceil_div(N, 256), (int)RB);
    if (wide) {
        colsum_partial_bf16x8_kernel<<<grid, 256, 0, s>>>((const bf16_t*)X, M, N, ldx, workspace);
    } else if (vec) {
        if (dt == VAW_F32) colsum_partial_kernel<float><<<grid, 256, 0, s>>>((const float*)X, M, N, ldx, workspace);
        else colsum_partial_kernel<bf16_t><<<grid, 256, 0, s>>>((const bf16_t*)X, M, N, ldx, workspace);
    } else {
        if (dt == VAW_F32) colsum_partial_scalar_kernel<float><<<grid, 256, 0, s>>>((const float*)X, M, N, ldx, workspace);
        else colsum_partial_scalar_kernel<bf16_t><<<grid, 256, 0, s>>>((const bf16_t*)X, M, N, ldx, workspace);
    }
    colsum_final_kernel<<<ceil_div(N, 32), 1024, 0, s>>>(workspace, RB, N, out, beta);
    VAW_CHECK_LAUNCH("colsum");
    return VAW_OK;
}
