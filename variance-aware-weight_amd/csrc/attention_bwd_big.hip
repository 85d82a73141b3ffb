// Attention backward (and, for the 96-wide head images, forward: attn_fwd_big below) for sequences that are multiples of 128 tokens up to 1024 (DiT-B/2 and DiT-XL/2: T = 256; the UNets' 16 x 16
// and 32 x 32 attention levels: T = 256 / 1024), head dims 40 .. 96: owner rows in registers, the other side streamed through a
// deep LDS-DMA ring in 32-row slices.  The default backward for these shapes since round 3 (attn_bwd_dq_mfma / attn_bwd_dkv_mfma of
// attention_mfma.hip keep the rest: T % 64, head dims up to 128).
//
// A wave OWNS 16 NT rows: their S- and dP-operand fragments (K and V rows for the key-owner kernel, Q and dO rows for the query-owner
// kernel) come straight from global memory into registers and stay there for the whole kernel next to the 16 NT x HD output
// accumulators; the other side streams through LDS in slices of 32 rows that all four waves share -- one fragment read feeds NT MFMAs.
//
//   MODE 0, key owner   (dK, dV):  S [q][key] = Q_s K_w^T and dP = dO_s V_w^T with the KEY on the lane; P and dS (accumulator
//                                  tiles, two query tiles packed = one k-step of 32) are the B operands of
//                                  dV^T[hd][key] += dO_s^T P  and  dK^T[hd][key] += Q_s^T dS  (contraction over the queries).
//   MODE 1, query owner (dQ, delta): S^T[key][q] = K_s Q_w^T, dP^T = V_s dO_w^T with the QUERY on the lane; dS^T feeds
//                                  dQ^T[hd][q] += K_s^T dS^T (contraction over the keys).  Also delta_i = rowsum(dO * O), from the
//                                  owner's dO fragments.
// Seven products for the pair (the minimum is five with dS handed across LDS; that kernel would have to sum dQ across waves).
// P = exp2(S c - lse log2 e), dS = P (dP - delta) with the scale applied once to the accumulated dK / dQ, bf16 operands, f32
// accumulation, k-steps of 32 rows in ascending order.  Outputs leave through LDS as whole rows; their column sums (the qkv bias
// gradient) are taken from the staged rows in a fixed order.
//
// What differs from the 16-row kernels, and what it is worth (tools/attn_bench.py, one box, interleaved; us per backward):
//   * slices arrive by LDS-DMA into a ring of FOUR buffers: three slices in flight while one is computed, counted vmcnt waits, one
//     raw barrier per slice (the 16-row kernels prefetch one 64-row block into registers and commit it after a barrier);
//   * every LDS read of the loop is inline asm: the compiler orders an ordinary LDS access behind ALL pending LDS-DMA with
//     s_waitcnt vmcnt(0), which would wait for the slices just put in flight; the transposed fragments of the next channel pair
//     are read under the MFMAs of the current one;
//   * everything a workgroup needs before its first MFMA -- first slices, owner fragments, row constants -- is requested at once.
//   NT = 2 (32 owner rows per wave, <= 256 registers, two workgroups per CU; VAW_ATTN_BWD_BIG=2, the default):
//       DiT-B/2 460 -> 375, DiT-XL/2 444 -> 405, ADM_64 32 x 32 2052 -> 1756, 16 x 16 270 -> 232, UNet_64 16 x 16 124 -> 108.
//   NT = 4 (64 owner rows per wave, 450 registers, ONE workgroup per CU; VAW_ATTN_BWD_BIG=1): half the LDS reads per MFMA, and at
//       PARITY with the 16-row kernels, not ahead (468 / 435 / 2057 / 321 / 106): with T = 256 and 1024 the kernel time fits
//       rounds x (F + slices x S) with S = 2.0 us per slice (96 MFMAs = 0.64 us of MFMA time) and F = 14 us of fixed cost per
//       workgroup (key owner, HD 96) -- a workgroup alone on its CU has nobody to cover its first memory latency, accumulator
//       set-up, two staged outputs and dispatch gap, nor the serial MFMA / softmax / LDS phases inside a slice (PMC: MFMA pipe busy
//       19 % of the wave time, VALU 32 %, waiting 33 %).  Two smaller workgroups cover each other; what the big one would need is
//       the slice loop as a hand-placed software pipeline and a persistent workgroup that prefetches the next pair's owner fragments.
// Where the default form's time goes (tools/attn_abl.sh: measurement builds -DBIG_ABL=n, results wrong, timing only; DiT-XL/2, DiT-B/2,
// ADM_64 32 x 32; base 400 / 350 / 1718 us): no slice barrier -1.5 %; no exp -1..-5 %; no LDS waits -1..-4 %; no barrier and no DMA wait
// -9 / -3 / -3 %; NO MFMAs at all -22 / -16 / -37 %; no slice DMA in the loop -23 / -6 / -12 %.  No single term dominates: the loop is
// ~335 vector instructions per slice and wave for 48 MFMAs (95 of them integer address arithmetic), and the fixed costs per workgroup
// weigh as much as the loop at T = 256.
// Reference: timm Attention's backward as autograd derives it (models/dit.py:126) / QKVAttention (models/unet.py:350-394).
#include "attention_mfma.h"

// ROWS x HD bf16 rows starting at g -> an LDS image with the layout of Img<HD> (all 4 waves cooperate, LDS-DMA)
template <int HD, int ROWS>
__device__ __forceinline__ void stage_rows(const bf16_t* __restrict__ g, int64_t stride_t, char* img, int wid, int lane, int hd) {
    constexpr int PCH = Img<HD>::PCH, TOTAL = ROWS * PCH, PIECES = (TOTAL + 63) / 64;
#pragma unroll
    for (int i = 0; i < (PIECES + 3) / 4; ++i) {
        const int piece = wid + 4 * i;
        if (PIECES % 4 != 0 && piece >= PIECES) break;          // uniform per wave
        const int idx = piece * 64 + lane;
        if (TOTAL % 64 == 0 || idx < TOTAL) {                   // (a last, partial piece: the other lanes write nothing)
            const int row = idx / PCH;
            const int chunk = (idx % PCH) ^ swz<HD>(row);
            const bf16_t* src = chunk * 8 < hd ? g + (int64_t)row * stride_t + chunk * 8
                                               : reinterpret_cast<const bf16_t*>(attn_zero_page);
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(img + piece * 1024), 16, 0, 0);
        }
    }
}

// column sums of ROWS staged rows (as stored: bf16) -> out_row[c], c < hd: four quarters, folded in ascending order
template <int HD, int ROWS>
__device__ __forceinline__ void cs_from_stage(const char* stage, int hd, float* red, float* out_row) {
    constexpr int QR = ROWS / 4;
    for (int idx = threadIdx.x; idx < 4 * HD; idx += 256) {
        const int qd = idx / HD, c = idx - qd * HD;
        if (c >= hd) continue;
        float sum = 0.f;
#pragma unroll 8
        for (int r = QR * qd; r < QR * qd + QR; ++r) sum += (float)*reinterpret_cast<const bf16_t*>(stage + out_off<HD>(r, c >> 3) + 2 * (c & 7));
        red[qd * HD + c] = sum;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < hd; c += 256) out_row[c] = ((red[c] + red[HD + c]) + red[2 * HD + c]) + red[3 * HD + c];
}

// LDS reads of the main loop go through inline asm: the compiler orders every ordinary LDS access behind ALL pending LDS-DMA with
// s_waitcnt vmcnt(0) (it cannot tell which image a read touches), which would wait for the slices just put in flight.  The asm reads
// carry no such wait; LDS_WAIT() is ours, before the first consumer (cdna_hip_programming.md rule 18).
__device__ __forceinline__ bf16x8 lds_rd128(unsigned addr) {
    bf16x8 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
__device__ __forceinline__ f32x4 lds_rd128f(unsigned addr) {
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
__device__ __forceinline__ bf16x4 lds_rd64tr(unsigned addr) {
    bf16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
#ifndef BIG_ABL
#define BIG_ABL 0
#endif
#if BIG_ABL == 3
#define LDS_WAIT() __builtin_amdgcn_sched_barrier(0)
#else
#define LDS_WAIT()                                          \
    do {                                                    \
        __builtin_amdgcn_sched_barrier(0);                  \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  \
        __builtin_amdgcn_sched_barrier(0);                  \
    } while (0)
#endif
#if BIG_ABL == 5
#undef MFMA
#define MFMA(a, b, c) big_fake_mfma(a, b, c)
__device__ __forceinline__ f32x4 big_fake_mfma(bf16x8 a, bf16x8 b, f32x4 c) {      // keeps the operands alive, one VALU op
    c[0] += (float)a[0] * (float)b[0];
    return c;
}
#endif
#if BIG_ABL == 2
#define BIG_EXP2(x) (x)
#else
#define BIG_EXP2(x) __builtin_amdgcn_exp2f(x)
#endif
// frag_rows / frag_cols_perm of attention_mfma.h on an image given by its LDS byte address
template <int HD>
__device__ __forceinline__ bf16x8 afrag_rows(unsigned img, int r0, int s, int lane) {
    return lds_rd128(img + (unsigned)img_off<HD>(r0 + (lane & 15), 4 * s + (lane >> 4)));
}
template <int HD>
__device__ __forceinline__ void afrag_cols_perm(unsigned img, int d0, int lane, bf16x4& lo, bf16x4& hi) {      // k = image rows 0 .. 31
    const int li = lane & 15, q = li >> 2, p = li & 3, g = lane >> 4;
    const int ch = (d0 >> 3) + (p >> 1);
    const int r_lo = 4 * g + q, r_hi = r_lo + 16;
    lo = lds_rd64tr(img + (unsigned)(img_off<HD>(r_lo, ch) + 8 * (p & 1)));
    hi = lds_rd64tr(img + (unsigned)(img_off<HD>(r_hi, ch) + 8 * (p & 1)));
}
__device__ __forceinline__ bf16x8 cat44(bf16x4 lo, bf16x4 hi) {
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
// counted wait for this wave's LDS-DMA: at most n of its youngest vector-memory instructions still in flight (n wave-uniform)
__device__ __forceinline__ void wait_vm(int n) {
    if (n >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (n >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (n >= 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int HD, int NT> struct BigLds {
    static constexpr int SIMG = 32 * Img<HD>::PITCH;                       // one 32-row slice image
    static constexpr int NB = 4;                                           // slice buffers: three slices in flight beside the one in use
    static constexpr int RING = NB * 2 * SIMG;
    static constexpr int WGR = 64 * NT;                                    // owner rows per workgroup (4 waves x NT tiles of 16)
    static constexpr int STAGE = WGR * (HD == 96 ? 208 : 2 * HD);         // the output rows, over the ring once the loop is done
    static constexpr int FRONT = RING > STAGE ? RING : STAGE;
    static int bytes(int T) { return FRONT + 2 * T * 4 + 4 * HD * 4; }
};

// NT = 16-row owner tiles per wave: 4 -> 64 rows per wave, one workgroup per CU (the whole register file); 2 -> 32 rows per wave,
// two workgroups per CU that cover each other's fixed costs and phases at twice the LDS reads per MFMA
template <int HD, int MODE, int NT>
__global__ void __launch_bounds__(256, NT == 2 ? 2 : 1)
attn_bwd_big(AttnMfmaArgs a, const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
             const bf16_t* __restrict__ o, const bf16_t* __restrict__ d_o, const float* __restrict__ lse, float* __restrict__ delta,
             bf16_t* __restrict__ out0, bf16_t* __restrict__ out1, float* __restrict__ cs_part, int64_t cs_ld) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KS = HD / 32, DT = HD / 16, SIMG = BigLds<HD, NT>::SIMG, WGR = BigLds<HD, NT>::WGR;
    constexpr int NB = BigLds<HD, NT>::NB, DEPTH = NB - 1;
    char* slices = smem;                                               // [NB buffers][X slice | Y slice]
    char* stage = smem;                                                // (the outputs are staged over the ring after the loop)
    float* lse_s = reinterpret_cast<float*>(smem + BigLds<HD, NT>::FRONT); // MODE 0: lse log2 e and delta of every query
    float* del_s = lse_s + a.T;
    float* red = del_s + a.T;
    const int lane = threadIdx.x & 63, g = lane >> 4, li = lane & 15;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int bh = blockIdx.y, b = bh / a.H, h = bh % a.H;
    const int64_t base = b * a.q_sb + h * a.q_sh, obase = b * a.o_sb + h * a.o_sh;
    const int own0 = blockIdx.x * WGR + 16 * NT * wid;                 // this wave's 16 NT owner rows (keys | queries)
    const float c2 = a.scale * 1.4426950408889634f;
    // streamed side: X_s feeds S, Y_s feeds dP
    const bf16_t* xs = MODE == 0 ? q + base : k + base;
    const bf16_t* ys = MODE == 0 ? d_o + obase : v + base;
    const int64_t xs_st = a.q_st, ys_st = MODE == 0 ? a.o_st : a.q_st;
    const int n_slices = a.T / 32;
    // this wave's LDS-DMA instructions per slice (two images; piece p of an image belongs to wave p % 4)
    constexpr int PIECES = (32 * Img<HD>::PCH + 63) / 64;
    const int dma_per_slice = 2 * ((PIECES - wid + 3) / 4);
    auto issue = [&](int sl) {
        char* nx = slices + (sl % NB) * 2 * SIMG;
        stage_rows<HD, 32>(xs + (int64_t)sl * 32 * xs_st, xs_st, nx, wid, lane, a.hd);
        stage_rows<HD, 32>(ys + (int64_t)sl * 32 * ys_st, ys_st, nx + SIMG, wid, lane, a.hd);
    };
    // Everything the workgroup needs before its first MFMA is requested at once -- the first slices, the owner fragments, the row
    // constants -- so that the workgroup (alone on its CU: nobody else covers its latencies) pays ONE memory latency, not three.
    for (int sl = 0; sl < DEPTH && sl < n_slices; ++sl) issue(sl);
    float lse_r[(MODE == 0) ? 4 : 1], del_r[(MODE == 0) ? 4 : 1];      // MODE 0: T <= 1024 row constants per thread, stored to LDS below
    if (MODE == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = threadIdx.x + 256 * j;
            lse_r[j] = i < a.T ? lse[(int64_t)bh * a.T + i] : 0.f;
            del_r[j] = i < a.T ? delta[(int64_t)bh * a.T + i] : 0.f;
        }
    }
    // owner side, straight from global memory into B-operand fragments: lane (li, g) holds channels 32 s + 8 g .. + 7 of row 16 t + li
    bf16x8 xf[NT][KS], yf[NT][KS];
    {
        const bf16_t* xo = MODE == 0 ? k + base : q + base;
        const bf16_t* yo = MODE == 0 ? v + base : d_o + obase;
        const int64_t yo_st = MODE == 0 ? a.q_st : a.o_st;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int ch = 32 * s + 8 * g;
                const int64_t row = own0 + 16 * t + li;
                const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
                xf[t][s] = ch < a.hd ? *reinterpret_cast<const bf16x8*>(xo + row * a.q_st + ch) : z;
                yf[t][s] = ch < a.hd ? *reinterpret_cast<const bf16x8*>(yo + row * yo_st + ch) : z;
            }
    }
    float lse_q[NT] = {}, del_q[NT] = {};           // MODE 1: of this lane's query in each of the 4 tiles
    if (MODE == 1) {
        // delta_i = sum_c dO[i][c] O[i][c]: dO is already here as yf (this lane: channels 32 s + 8 g .. + 7 of row 16 t + li)
        bf16x8 of[NT][KS];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int ch = 32 * s + 8 * g;
                const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
                of[t][s] = ch < a.hd ? *reinterpret_cast<const bf16x8*>(o + obase + (int64_t)(own0 + 16 * t + li) * a.o_st + ch) : z;
            }
#pragma unroll
        for (int t = 0; t < NT; ++t) lse_q[t] = lse[(int64_t)bh * a.T + own0 + 16 * t + li] * 1.4426950408889634f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float dl = 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int j = 0; j < 8; ++j) dl += (float)yf[t][s][j] * (float)of[t][s][j];
            dl = group_sum(dl);
            del_q[t] = dl;
            if (g == 0) delta[(int64_t)bh * a.T + own0 + 16 * t + li] = dl;
        }
    } else {
        // (ordinary LDS stores: the compiler puts them behind every pending LDS-DMA -- the slices requested above, which the loop needs anyway)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = threadIdx.x + 256 * j;
            if (i < a.T) {
                lse_s[i] = lse_r[j] * 1.4426950408889634f;
                del_s[i] = del_r[j];
            }
        }
    }
    f32x4 acc0[NT][DT], acc1[MODE == 0 ? NT : 1][DT];                   // MODE 0: dK^T, dV^T per key tile; MODE 1: dQ^T per query tile
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            acc0[t][dt] = f32x4{0, 0, 0, 0};
            if (MODE == 0) acc1[t][dt] = f32x4{0, 0, 0, 0};
        }
    const unsigned ring = (unsigned)(uintptr_t)(lds_ptr_t)slices;
    const unsigned lse_a = (unsigned)(uintptr_t)(lds_ptr_t)lse_s, del_a = (unsigned)(uintptr_t)(lds_ptr_t)del_s;
    if (MODE == 0) __syncthreads();                      // lse_s / del_s are complete
    for (int sl = 0; sl < n_slices; ++sl) {
        const unsigned ximg = ring + (unsigned)((sl % NB) * 2 * SIMG), yimg = ximg + SIMG;
        {
            const int last = sl + DEPTH - 1 < n_slices - 1 ? sl + DEPTH - 1 : n_slices - 1;      // youngest slice issued so far
#if BIG_ABL != 4
            wait_vm((last - sl) * dma_per_slice);        // this wave's pieces of slice sl have landed ...
#endif
#if BIG_ABL != 1 && BIG_ABL != 4                         // (measurement builds, -DBIG_ABL=n: 1 no barrier, 2 no exp, 3 no LDS waits, 4 no
                                                         //  barrier and no DMA wait, 5 no MFMAs, 6 no slice DMA in the loop: wrong results)
            __builtin_amdgcn_s_barrier();                // ... everybody's have; and everybody is done with slice sl - 1
#endif
#if BIG_ABL != 6
            if (sl + DEPTH < n_slices) issue(sl + DEPTH);                                       // into the buffer slice sl - 1 used
#endif
        }
        if (MODE == 0) {
            bf16x8 xa[2][KS], ya[2][KS];
            f32x4 L4[2], D4[2];
#pragma unroll
            for (int it = 0; it < 2; ++it) {
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    xa[it][s] = afrag_rows<HD>(ximg, 16 * it, s, lane);
                    ya[it][s] = afrag_rows<HD>(yimg, 16 * it, s, lane);
                }
                const unsigned i0 = 4u * (unsigned)(32 * sl + 16 * it + 4 * g);
                L4[it] = lds_rd128f(lse_a + i0);
                D4[it] = lds_rd128f(del_a + i0);
            }
            LDS_WAIT();
            // the first pair of transposed fragments flies under the S / dP products; every later pair under the products before it
            bf16x4 yl[2][2], yh[2][2], xl[2][2], xh[2][2];           // [ping-pong][channel tile of the pair]
            auto rd_pair = [&](int dh, int pp) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    afrag_cols_perm<HD>(yimg, 16 * (dh + u), lane, yl[pp][u], yh[pp][u]);
                    afrag_cols_perm<HD>(ximg, 16 * (dh + u), lane, xl[pp][u], xh[pp][u]);
                }
            };
            rd_pair(0, 0);
            const f32x4 zero4 = {0, 0, 0, 0};
            f32x4 p[2][NT], ds[2][NT];
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                f32x4 c[NT], d[NT];
#pragma unroll
                for (int s = 0; s < KS; ++s)
#pragma unroll
                    for (int jt = 0; jt < NT; ++jt) {
                        c[jt] = MFMA(xa[it][s], xf[jt][s], s == 0 ? zero4 : c[jt]);          // S [query 4g + r][key li]
                        d[jt] = MFMA(ya[it][s], yf[jt][s], s == 0 ? zero4 : d[jt]);          // dP
                    }
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int jt = 0; jt < NT; ++jt) {
                        const float pr = BIG_EXP2(c[jt][r] * c2 - L4[it][r]);
                        p[it][jt][r] = pr;
                        ds[it][jt][r] = pr * (d[jt][r] - D4[it][r]);      // (x scale: once, on the accumulated dK^T)
                    }
            }
            bf16x8 pf[NT], sf[NT];
#pragma unroll
            for (int jt = 0; jt < NT; ++jt) {
                pf[jt] = pack_acc(p[0][jt], p[1][jt]);
                sf[jt] = pack_acc(ds[0][jt], ds[1][jt]);
            }
#pragma unroll
            for (int dh = 0; dh < DT; dh += 2) {             // transposed fragments two channel tiles at a time
                const int pp = (dh >> 1) & 1;
                LDS_WAIT();
                if (dh + 2 < DT) rd_pair(dh + 2, pp ^ 1);
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const bf16x8 yt = cat44(yl[pp][u], yh[pp][u]), xt = cat44(xl[pp][u], xh[pp][u]);
#pragma unroll
                    for (int jt = 0; jt < NT; ++jt) {
                        acc1[jt][dh + u] = MFMA(yt, pf[jt], acc1[jt][dh + u]);    // dV^T [channel][key]
                        acc0[jt][dh + u] = MFMA(xt, sf[jt], acc0[jt][dh + u]);    // dK^T / scale
                    }
                }
            }
        } else {
            bf16x8 xa[2][KS], ya[2][KS];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    xa[kt][s] = afrag_rows<HD>(ximg, 16 * kt, s, lane);
                    ya[kt][s] = afrag_rows<HD>(yimg, 16 * kt, s, lane);
                }
            LDS_WAIT();
            bf16x4 xl[2][2], xh[2][2];
            auto rd_pair = [&](int dh, int pp) {
#pragma unroll
                for (int u = 0; u < 2; ++u) afrag_cols_perm<HD>(ximg, 16 * (dh + u), lane, xl[pp][u], xh[pp][u]);
            };
            rd_pair(0, 0);
            const f32x4 zero4 = {0, 0, 0, 0};
            f32x4 ds[2][NT];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                f32x4 c[NT], d[NT];
#pragma unroll
                for (int s = 0; s < KS; ++s)
#pragma unroll
                    for (int qt = 0; qt < NT; ++qt) {
                        c[qt] = MFMA(xa[kt][s], xf[qt][s], s == 0 ? zero4 : c[qt]);          // S^T [key 4g + r][query li]
                        d[qt] = MFMA(ya[kt][s], yf[qt][s], s == 0 ? zero4 : d[qt]);          // dP^T
                    }
#pragma unroll
                for (int qt = 0; qt < NT; ++qt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        ds[kt][qt][r] = BIG_EXP2(c[qt][r] * c2 - lse_q[qt]) * (d[qt][r] - del_q[qt]);      // (x scale: on dQ^T)
            }
            bf16x8 sf[NT];
#pragma unroll
            for (int qt = 0; qt < NT; ++qt) sf[qt] = pack_acc(ds[0][qt], ds[1][qt]);
#pragma unroll
            for (int dh = 0; dh < DT; dh += 2) {
                const int pp = (dh >> 1) & 1;
                LDS_WAIT();
                if (dh + 2 < DT) rd_pair(dh + 2, pp ^ 1);
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const bf16x8 xt = cat44(xl[pp][u], xh[pp][u]);
#pragma unroll
                    for (int qt = 0; qt < NT; ++qt) acc0[qt][dh + u] = MFMA(xt, sf[qt], acc0[qt][dh + u]);    // dQ^T / scale
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    // ---- outputs: 256 rows per workgroup through the staging tile, whole rows out, column sums from the staged values ----
    const int64_t orow0 = (int64_t)blockIdx.x * WGR;
    float* cs_row = cs_part ? cs_part + ((int64_t)b * gridDim.x + blockIdx.x) * cs_ld : nullptr;
    const int Hhd = a.H * a.hd;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) acc0[t][dt] *= a.scale;      // dS was accumulated without its scale factor
#pragma unroll
    for (int which = 0; which < (MODE == 0 ? 2 : 1); ++which) {
        __syncthreads();                                 // (first pass: the last slice's reads; second: the flush before)
#pragma unroll
        for (int t = 0; t < NT; ++t) out_stage16<HD, DT>(stage, 16 * NT * wid + 16 * t, which == 0 ? acc0[t] : acc1[MODE == 0 ? t : 0], lane);
        __syncthreads();
        bf16_t* dst = (which == 0 ? out0 : out1) + base + orow0 * a.q_st;
        out_flush<HD>(stage, WGR, dst, a.q_st, a.hd);
        if (cs_row) {
            // packed qkv column order [3][H][hd]: dq = 0, dk = 1, dv = 2
            const int col = (MODE == 1 ? 0 : 1 + which) * Hhd + h * a.hd;
            cs_from_stage<HD, WGR>(stage, a.hd, red, cs_row + col);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// forward in the same form: a wave owns 16 NT queries (Q fragments in registers, O^T accumulators), K and V stream through the ring
// in 32-key slices; online softmax per query with the running maximum on the lane (base-2 domain, deferred rescale as in
// attn_fwd_mfma: the maximum only moves when a slice exceeds it by more than 2^6).  S^T[key][q] = K_s Q_w^T, O^T[hd][q] += V_s^T P^T.
// ------------------------------------------------------------------------------------------------
#ifndef ATTN_PROF
#define ATTN_PROF 0
#endif
#if ATTN_PROF
__device__ unsigned long long attn_prof_buf[1024];
extern "C" int vaw_debug_attn_prof(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(attn_prof_buf), sizeof(unsigned long long) * n);
}
#define ATTN_STAMP(slot)                                                                                                     \
    do {                                                                                                                     \
        if (threadIdx.x == 0 && blockIdx.x == 1 && blockIdx.y == 777 && prof_n < 1000)                                       \
            attn_prof_buf[prof_n++] = ((unsigned long long)(slot) << 56) | (wall_clock64() & 0xffffffffffffffull);           \
    } while (0)
#else
#define ATTN_STAMP(slot) do {} while (0)
#endif
// the forward's own ring depth / occupancy target (A/B: -DATTN_FWD_NB=3 -DATTN_FWD_OCC=4 = four workgroups per CU on a 3-deep ring)
#ifndef ATTN_FWD_NB
#define ATTN_FWD_NB 4
#endif
#ifndef ATTN_FWD_OCC
#define ATTN_FWD_OCC 2
#endif
template <int HD, int NT> struct FwdLds {
    static constexpr int SIMG = BigLds<HD, NT>::SIMG, NB = ATTN_FWD_NB, RING = NB * 2 * SIMG, WGR = BigLds<HD, NT>::WGR;
    static constexpr int STAGE = BigLds<HD, NT>::STAGE, FRONT = RING > STAGE ? RING : STAGE;
};
template <int HD, int NT>
__global__ void __launch_bounds__(256, NT == 2 ? ATTN_FWD_OCC : 1)
attn_fwd_big(AttnMfmaArgs a, const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
             bf16_t* __restrict__ o, float* __restrict__ lse) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KS = HD / 32, DT = HD / 16, SIMG = FwdLds<HD, NT>::SIMG, WGR = FwdLds<HD, NT>::WGR;
    constexpr int NB = FwdLds<HD, NT>::NB, DEPTH = NB - 1;
#if ATTN_PROF
    int prof_n = 0;
#endif
    ATTN_STAMP(1);
    char* slices = smem;
    char* stage = smem;
    const int lane = threadIdx.x & 63, g = lane >> 4, li = lane & 15;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int bh = blockIdx.y, b = bh / a.H, h = bh % a.H;
    const int64_t base = b * a.q_sb + h * a.q_sh, obase = b * a.o_sb + h * a.o_sh;
    const int own0 = blockIdx.x * WGR + 16 * NT * wid;
    const float c2 = a.scale * 1.4426950408889634f;
    const bf16_t* xs = k + base;
    const bf16_t* ys = v + base;
    const int n_slices = a.T / 32;
    constexpr int PIECES = (32 * Img<HD>::PCH + 63) / 64;
    const int dma_per_slice = 2 * ((PIECES - wid + 3) / 4);
    auto issue = [&](int sl) {
        char* nx = slices + (sl % NB) * 2 * SIMG;
        stage_rows<HD, 32>(xs + (int64_t)sl * 32 * a.q_st, a.q_st, nx, wid, lane, a.hd);
        stage_rows<HD, 32>(ys + (int64_t)sl * 32 * a.q_st, a.q_st, nx + SIMG, wid, lane, a.hd);
    };
    for (int sl = 0; sl < DEPTH && sl < n_slices; ++sl) issue(sl);
    bf16x8 qf[NT][KS];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int ch = 32 * s + 8 * g;
            const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
            qf[t][s] = ch < a.hd ? *reinterpret_cast<const bf16x8*>(q + base + (int64_t)(own0 + 16 * t + li) * a.q_st + ch) : z;
        }
    f32x4 ot[NT][DT];
    float m[NT], l[NT];                  // running maximum (whole query: equal on its 4 lanes), this LANE's share of the running sum
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        m[t] = -INFINITY;
        l[t] = 0.f;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) ot[t][dt] = f32x4{0, 0, 0, 0};
    }
    const unsigned ring = (unsigned)(uintptr_t)(lds_ptr_t)slices;
    for (int sl = 0; sl < n_slices; ++sl) {
        const unsigned ximg = ring + (unsigned)((sl % NB) * 2 * SIMG), yimg = ximg + SIMG;
        {
            const int last = sl + DEPTH - 1 < n_slices - 1 ? sl + DEPTH - 1 : n_slices - 1;
            wait_vm((last - sl) * dma_per_slice);
            __builtin_amdgcn_s_barrier();
            ATTN_STAMP(2);
            if (sl + DEPTH < n_slices) issue(sl + DEPTH);
        }
        bf16x8 xa[2][KS];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int s = 0; s < KS; ++s) xa[kt][s] = afrag_rows<HD>(ximg, 16 * kt, s, lane);
        LDS_WAIT();
        bf16x4 yl[2][2], yh[2][2];
        auto rd_pair = [&](int dh, int pp) {
#pragma unroll
            for (int u = 0; u < 2; ++u) afrag_cols_perm<HD>(yimg, 16 * (dh + u), lane, yl[pp][u], yh[pp][u]);
        };
        rd_pair(0, 0);
        const f32x4 zero4 = {0, 0, 0, 0};
        f32x4 c[2][NT];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int qt = 0; qt < NT; ++qt) c[kt][qt] = MFMA(xa[kt][s], qf[qt][s], s == 0 ? zero4 : c[kt][qt]);      // S^T [key 4g + r][query li]
        ATTN_STAMP(3);
        bf16x8 pf[NT];
        bool any_rescale = false;
        float alpha[NT];
#pragma unroll
        for (int qt = 0; qt < NT; ++qt) {
            float bm = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) bm = fmaxf(bm, c[kt][qt][r]);
            float m_new = fmaxf(m[qt], group_max(bm) * c2);
            if (m_new - m[qt] <= 6.f) m_new = m[qt];            // (first slice: m = -inf -> always taken over)
            alpha[qt] = __builtin_amdgcn_exp2f(m[qt] - m_new);   // exp2(-inf) = 0 on l = 0, ot = 0; 1 when the maximum stayed
            float ps = 0.f;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    c[kt][qt][r] = __builtin_amdgcn_exp2f(c[kt][qt][r] * c2 - m_new);
                    ps += c[kt][qt][r];
                }
            l[qt] = l[qt] * alpha[qt] + ps;
            m[qt] = m_new;
            pf[qt] = pack_acc(c[0][qt], c[1][qt]);
            any_rescale = any_rescale || alpha[qt] != 1.f;
        }
        if (!__all(!any_rescale)) {
#pragma unroll
            for (int qt = 0; qt < NT; ++qt)
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) ot[qt][dt] *= alpha[qt];
        }
        ATTN_STAMP(4);
#pragma unroll
        for (int dh = 0; dh < DT; dh += 2) {
            const int pp = (dh >> 1) & 1;
            LDS_WAIT();
            if (dh + 2 < DT) rd_pair(dh + 2, pp ^ 1);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const bf16x8 yt = cat44(yl[pp][u], yh[pp][u]);
#pragma unroll
                for (int qt = 0; qt < NT; ++qt) ot[qt][dh + u] = MFMA(yt, pf[qt], ot[qt][dh + u]);    // O^T [channel][query]
            }
        }
    }
    ATTN_STAMP(5);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();                                     // the last slice's reads: O is staged over the ring
    ATTN_STAMP(6);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const float lt = group_sum(l[t]);
        const float inv = 1.f / lt;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) ot[t][dt] *= inv;
        out_stage16<HD, DT>(stage, 16 * NT * wid + 16 * t, ot[t], lane);
        if (g == 0) lse[(int64_t)bh * a.T + own0 + 16 * t + li] = m[t] * 0.6931471805599453f + __logf(lt);   // back to the natural log
    }
    __syncthreads();
    ATTN_STAMP(7);
    out_flush<HD>(stage, WGR, o + obase + (int64_t)blockIdx.x * WGR * a.o_st, a.o_st, a.hd);
    ATTN_STAMP(8);
}

static AttnMfmaArgs mk_args_big(const vaw_attn_desc* d) {
    AttnMfmaArgs a{d->B, d->H, d->T, d->q_sb, d->q_sh, d->q_st, d->o_sb, d->o_sh, d->o_st, d->scale, d->hd};
    return a;
}

// true when the shape is taken (both launches enqueued); cs_part / cs_rows_out as in vaw_attn_bwd_mfma
template <int HD, int NT>
static void big_go(const AttnMfmaArgs& a, const vaw_attn_desc* d, const void* q, const void* k, const void* v, const void* o, const void* d_o,
                   const float* lse, float* delta, void* dq, void* dk, void* dv, hipStream_t s, float* cs_part, int64_t cs_ld) {
    const int lds = BigLds<HD, NT>::bytes(d->T);
    dim3 grid(d->T / BigLds<HD, NT>::WGR, d->B * d->H);
    static bool attr_done = false;                       // (once per instantiation: the limit for the longest sequence taken, T = 1024)
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)attn_bwd_big<HD, 1, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, BigLds<HD, NT>::bytes(1024));
        (void)hipFuncSetAttribute((const void*)attn_bwd_big<HD, 0, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, BigLds<HD, NT>::bytes(1024));
        attr_done = true;
    }
    attn_bwd_big<HD, 1, NT><<<grid, 256, lds, s>>>(a, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (const bf16_t*)o, (const bf16_t*)d_o,
                                                   lse, delta, (bf16_t*)dq, nullptr, cs_part, cs_ld);
    attn_bwd_big<HD, 0, NT><<<grid, 256, lds, s>>>(a, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (const bf16_t*)o, (const bf16_t*)d_o,
                                                   lse, delta, (bf16_t*)dk, (bf16_t*)dv, cs_part, cs_ld);
}

bool vaw_attn_bwd_big(const vaw_attn_desc* d, const void* q, const void* k, const void* v, const void* o, const void* d_o,
                      const float* lse, float* delta, void* dq, void* dk, void* dv, hipStream_t s, float* cs_part, int64_t* cs_rows_out) {
    // VAW_ATTN_BWD_BIG: 2 (default) = 32 owner rows per wave, two workgroups per CU (T % 128 == 0); 1 = 64 owner rows per wave, one
    // workgroup per CU (T % 256 == 0; the experiment of the header); 0 = the 16-row kernels of attention_mfma.hip.  Read per call.
    const char* e = getenv("VAW_ATTN_BWD_BIG");
    const int on = e ? atoi(e) : 2;
    const int wgr = on == 2 ? 128 : 256;
    if ((on != 1 && on != 2) || d->T % wgr != 0 || d->T > 1024 || d->hd <= 32 || d->hd > 96) return false;
    const AttnMfmaArgs a = mk_args_big(d);
    const int64_t cs_ld = 3LL * d->H * d->hd;
    if (d->hd <= 64) { if (on == 2) big_go<64, 2>(a, d, q, k, v, o, d_o, lse, delta, dq, dk, dv, s, cs_part, cs_ld); else big_go<64, 4>(a, d, q, k, v, o, d_o, lse, delta, dq, dk, dv, s, cs_part, cs_ld); }
    else { if (on == 2) big_go<96, 2>(a, d, q, k, v, o, d_o, lse, delta, dq, dk, dv, s, cs_part, cs_ld); else big_go<96, 4>(a, d, q, k, v, o, d_o, lse, delta, dq, dk, dv, s, cs_part, cs_ld); }
    if (cs_rows_out) *cs_rows_out = (int64_t)d->B * (d->T / wgr);
    return true;
}

template <int HD, int NT>
static void big_fwd_go(const AttnMfmaArgs& a, const vaw_attn_desc* d, const void* q, const void* k, const void* v, void* o, float* lse, hipStream_t s) {
    const int lds = FwdLds<HD, NT>::FRONT;
    dim3 grid(d->T / FwdLds<HD, NT>::WGR, d->B * d->H);
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)attn_fwd_big<HD, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_done = true;
    }
    attn_fwd_big<HD, NT><<<grid, 256, lds, s>>>(a, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)o, lse);
}

// The owner-rows forward (T % 128 == 0).  Measured against attn_fwd_mfma (tools/attn_bench.py, interleaved): the 96-wide images win
// (DiT-XL/2 138.7 -> 120.8 us, UNet_64 16 x 16 39.2 -> 36.6), the 64-wide ones -- where attn_fwd_mfma double-buffers K / V by LDS-DMA and
// keeps three workgroups per CU -- lose 2-4 % (DiT-B/2 103.0 -> 105.5, ADM_64 32 x 32 641 -> 668).  So: head dims 72 .. 96 by default;
// VAW_ATTN_FWD_BIG=1 all of 40 .. 96, =0 none.  Read per call.
bool vaw_attn_fwd_big(const vaw_attn_desc* d, const void* q, const void* k, const void* v, void* o, float* lse, hipStream_t s) {
    const char* e = getenv("VAW_ATTN_FWD_BIG");
    const int on = e ? atoi(e) : 2;
    const int hd_lo = on == 1 ? 32 : 64;
    if (on == 0 || d->T % 128 != 0 || d->hd <= hd_lo || d->hd > 96) return false;
    const AttnMfmaArgs a = mk_args_big(d);
    if (d->hd <= 64) big_fwd_go<64, 2>(a, d, q, k, v, o, lse, s); else big_fwd_go<96, 2>(a, d, q, k, v, o, lse, s);
    return true;
}
