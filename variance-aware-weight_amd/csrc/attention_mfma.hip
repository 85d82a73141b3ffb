// bf16 MFMA attention for token-major q/k/v rows (DiT's timm Attention AND the UNet's QKVAttention once its
// activations are NHWC): head dim up to 128 in steps of 8 (padded with zero columns to HD in {32, 64, 96, 128}), any sequence length T that is a multiple of 64.
//
// Everything is computed TRANSPOSED so that no probability tile ever crosses lanes or LDS:
//   S^T[key][query] = K . Q^T  lands in the 16x16 accumulator layout with the QUERY on the lane (col = l&15) and
//   the keys in registers (row = 4(l>>4)+r); softmax statistics are then an in-lane loop plus two shuffles
//   (xor 16, 32), and bf16(P^T) is ALREADY the B operand of the next product O^T = V^T . P^T
//   (cdna_hip_programming.md §3 "An accumulator tile as the next MFMA's operand": the k order of such an operand
//   is permuted -- k slot (g, jj) holds accumulator row 4g+jj of tile 2s (jj<4) or tile 2s+1 (jj>=4) -- so the A
//   operand V^T is fetched with the same permutation by ds_read_b64_tr_b16).
// Keys (forward, dQ) or queries (dK/dV) stream through LDS in blocks of 64 rows; the forward keeps a running
// (max, sum) per query (online softmax), the backward recomputes P from the saved row log-sum-exp in BOTH
// orientations (S^T for dQ, S for dK/dV) -- two extra products per tile buy kernels with no transposes, no
// atomics and no T x T tensor in memory.  delta_i = rowsum(dO * O).
//
// LDS: every operand block is a [64 rows][HD] bf16 image filled by LDS-DMA (global_load_lds_dwordx4) with the
// 16-byte chunk XOR-swizzled on the SOURCE side so that the ds_read_b128 row reads are conflict-free
// (HD 32/64/128).  HD 96 (DiT-XL's 72-wide heads, the UNets' 96): an XOR cannot stay inside a 12-chunk row, so the row pitch is
// padded to 13 chunks (208 bytes) instead: 52 dwords per row puts 16 consecutive rows on 16 distinct 4-bank groups, which makes
// both the ds_read_b128 row reads and the transposed column reads conflict-free (the linear 192-byte rows were 4-way conflicted);
// the pad chunk is filled from the zero page by the same DMA instruction stream (13 wave-instructions per image instead of 12).
#include "attention_mfma.h"

// ------------------------------------------------------------------------------------------------
// forward: grid (T/64, B*H), 4 waves x 16 queries; keys stream in blocks of 64 with an online softmax
// ------------------------------------------------------------------------------------------------
// QG = 16-query groups per wave: 1 -> 64 queries per workgroup, 2 -> 128 (T % 128 == 0): every K / V fragment read from
// LDS then feeds two MFMAs, and the per-key-block barrier + DMA wait is paid once per 128 queries.
// DB: K / V blocks double-buffered (the next block streams in while this one is consumed; one barrier per block):
// +10..19 % for head dims <= 64; the wider images lose a resident workgroup to the extra LDS and keep one buffer.
#ifndef ATTN64_XCD
#define ATTN64_XCD 0        // 1: each XCD takes a contiguous run of (sample, head) pairs -- measured level to worse (fwd 21.9 -> 24.8 us)
#endif
template <int HD, int QG, bool DB = (HD <= 64)>
__global__ void __launch_bounds__(256)
attn_fwd_mfma(AttnMfmaArgs a, const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
              bf16_t* __restrict__ o, float* __restrict__ lse) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // QG + 2 (+ 2 if DB) images of 64 x HD bf16
    constexpr int KS = HD / 32, DT = HD / 16, IMG = Img<HD>::BYTES;
    char* qimg = smem;
    char* kv = smem + QG * IMG;                                   // [1 or 2 buffers][K image | V image]
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // one query block per (sample, head) (T = 64 QG): workgroups follow blockIdx.y round-robin over the XCDs; each XCD takes a
    // contiguous run of pairs instead (see attn_bwd_t64_mfma)
    int bh = blockIdx.y;
    if (ATTN64_XCD && gridDim.x == 1 && (gridDim.y & 7) == 0) bh = (blockIdx.y & 7) * (gridDim.y >> 3) + (blockIdx.y >> 3);
    const int b = bh / a.H, h = bh % a.H;
    const int qb = blockIdx.x * 64 * QG;
    const int64_t base = b * a.q_sb + h * a.q_sh;
#pragma unroll
    for (int u = 0; u < QG; ++u) stage_block<HD>(q + base + (int64_t)(qb + 64 * u) * a.q_st, a.q_st, qimg + u * IMG, wid, lane, a.hd);
    bf16x8 qf[QG][KS];
    f32x4 ot[QG][DT];
    float m[QG], l[QG];
    const float c2 = a.scale * 1.4426950408889634f;              // scores to the base-2 exponent domain
#pragma unroll
    for (int u = 0; u < QG; ++u) {
        m[u] = -INFINITY;
        l[u] = 0.f;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) ot[u][dt] = f32x4{0, 0, 0, 0};
    }
    BlkRegs<HD> kr, vr;                                           // !DB: the NEXT key block, in flight in registers
    if (DB) {
        stage_block<HD>(k + base, a.q_st, kv, wid, lane, a.hd);
        stage_block<HD>(v + base, a.q_st, kv + IMG, wid, lane, a.hd);
    } else {
        blk_prefetch<HD>(kr, k + base, a.q_st, a.hd);
        blk_prefetch<HD>(vr, v + base, a.q_st, a.hd);
    }
    for (int kb = 0, buf = 0; kb < a.T; kb += 64, buf ^= (DB ? 1 : 0)) {
        const char* kimg = kv + buf * 2 * IMG;
        const char* vimg = kimg + IMG;
        if (DB) {
            DMA_WAIT_SYNC();                                 // this key block has landed; nobody still reads the other buffer
            if (kb + 64 < a.T) {
                stage_block<HD>(k + base + (int64_t)(kb + 64) * a.q_st, a.q_st, kv + (buf ^ 1) * 2 * IMG, wid, lane, a.hd);
                stage_block<HD>(v + base + (int64_t)(kb + 64) * a.q_st, a.q_st, kv + (buf ^ 1) * 2 * IMG + IMG, wid, lane, a.hd);
            }
        } else {
            __syncthreads();                                 // previous block's reads of the K / V images are done
            blk_commit<HD>(kr, kv);
            blk_commit<HD>(vr, kv + IMG);
            DMA_WAIT_SYNC();                                 // (the Q images' LDS-DMA, first block; then a plain barrier)
            if (kb + 64 < a.T) {                             // the next block's loads fly while this one is computed
                blk_prefetch<HD>(kr, k + base + (int64_t)(kb + 64) * a.q_st, a.q_st, a.hd);
                blk_prefetch<HD>(vr, v + base + (int64_t)(kb + 64) * a.q_st, a.q_st, a.hd);
            }
        }
        if (kb == 0) {
#pragma unroll
            for (int u = 0; u < QG; ++u)
#pragma unroll
                for (int s = 0; s < KS; ++s) qf[u][s] = frag_rows<HD>(qimg, 16 * (QG * wid + u), s, lane);
        }
        f32x4 st[QG][4];
        float bm[QG];
#pragma unroll
        for (int u = 0; u < QG; ++u) bm[u] = -INFINITY;
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            f32x4 c[QG];
#pragma unroll
            for (int u = 0; u < QG; ++u) c[u] = f32x4{0, 0, 0, 0};
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const bf16x8 kf = frag_rows<HD>(kimg, 16 * jt, s, lane);
#pragma unroll
                for (int u = 0; u < QG; ++u) c[u] = MFMA(kf, qf[u][s], c[u]);
            }
#pragma unroll
            for (int u = 0; u < QG; ++u) {
                st[u][jt] = c[u];
#pragma unroll
                for (int r = 0; r < 4; ++r) bm[u] = fmaxf(bm[u], c[u][r]);
            }
        }
#pragma unroll
        for (int u = 0; u < QG; ++u) {
            // base-2 softmax: p = exp2(s c - m), c = scale log2(e) -- one fma + v_exp_f32 per score.  The running maximum only
            // moves when the block's maximum exceeds it by more than 2^6 (cdna_hip_programming.md T13: p stays <= 64, exact in
            // f32 and at bf16's relative precision); when no query of the wave moved it, the 4 DT-register rescale of O is skipped
            float m_new = fmaxf(m[u], group_max(bm[u]) * c2);
            if (m_new - m[u] <= 6.f) m_new = m[u];               // (first block: m = -inf -> always taken over)
            const float alpha = __builtin_amdgcn_exp2f(m[u] - m_new);   // exp2(-inf) = 0 on l = 0, ot = 0; 1 when the maximum stayed
            float ps = 0.f;
#pragma unroll
            for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    st[u][jt][r] = __builtin_amdgcn_exp2f(st[u][jt][r] * c2 - m_new);
                    ps += st[u][jt][r];
                }
            l[u] = l[u] * alpha + group_sum(ps);
            m[u] = m_new;
            if (!__all(alpha == 1.f)) {
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) ot[u][dt] *= alpha;
            }
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            bf16x8 pf[QG];
#pragma unroll
            for (int u = 0; u < QG; ++u) pf[u] = pack_acc(st[u][2 * s2], st[u][2 * s2 + 1]);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const bf16x8 vf = frag_cols_perm<HD>(vimg, 16 * dt, 32 * s2, lane);
#pragma unroll
                for (int u = 0; u < QG; ++u) ot[u][dt] = MFMA(vf, pf[u], ot[u][dt]);
            }
        }
    }
    __syncthreads();                                              // every wave is done with the Q / K / V images: O is staged over Q
#pragma unroll
    for (int u = 0; u < QG; ++u) {
        const float inv = 1.f / l[u];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) ot[u][dt] *= inv;
        out_stage16<HD, DT>(qimg, 16 * (QG * wid + u), ot[u], lane);
        const int qi = qb + 16 * (QG * wid + u) + (lane & 15);
        if ((lane >> 4) == 0) lse[(int64_t)bh * a.T + qi] = m[u] * 0.6931471805599453f + __logf(l[u]);   // back to the natural log
    }
    __syncthreads();
    out_flush<HD>(qimg, 64 * QG, o + b * a.o_sb + h * a.o_sh + (int64_t)qb * a.o_st, a.o_st, a.hd);
}

// ------------------------------------------------------------------------------------------------
// backward, query-major: grid (T/(64 G), B*H); delta and dQ for 64 G queries, keys stream.
// G = 2 (T % 128 == 0, 96-wide head images): every staged K / V block and every fragment read from it serves two query
// groups per wave.
// ------------------------------------------------------------------------------------------------
template <int HD, int G>
__global__ void __launch_bounds__(256)
attn_bwd_dq_mfma(AttnMfmaArgs a, const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                 const bf16_t* __restrict__ o, const bf16_t* __restrict__ d_o, const float* __restrict__ lse,
                 float* __restrict__ delta_out, bf16_t* __restrict__ dq, float* __restrict__ cs_part = nullptr, int64_t cs_ld = 0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 G + 2 images of 64 x HD bf16
    constexpr int KS = HD / 32, DT = HD / 16, IMG = Img<HD>::BYTES;
    char* qimg = smem;                       // G images
    char* gimg = qimg + G * IMG;             // G images
    char* kimg = gimg + G * IMG;
    char* vimg = kimg + IMG;
    const float c2 = a.scale * 1.4426950408889634f;              // scores to the base-2 exponent domain
    const int lane = threadIdx.x & 63, g = lane >> 4, li = lane & 15;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int bh = blockIdx.y, b = bh / a.H, h = bh % a.H;
    const int qb = blockIdx.x * 64 * G;
    const int64_t base = b * a.q_sb + h * a.q_sh, obase = b * a.o_sb + h * a.o_sh;
#pragma unroll
    for (int u = 0; u < G; ++u) {
        stage_block<HD>(q + base + (int64_t)(qb + 64 * u) * a.q_st, a.q_st, qimg + u * IMG, wid, lane, a.hd);
        stage_block<HD>(d_o + obase + (int64_t)(qb + 64 * u) * a.o_st, a.o_st, gimg + u * IMG, wid, lane, a.hd);
    }
    int qi[G];
    float dl[G], li_lse[G];
#pragma unroll
    for (int u = 0; u < G; ++u) {
        qi[u] = qb + 64 * u + 16 * wid + li;
        // delta_i = sum_d dO[i,d] * O[i,d]: lane (li, g) takes a quarter of the row
        float t = 0.f;
        const bf16_t* gp = d_o + obase + (int64_t)qi[u] * a.o_st + g * (HD / 4);
        const bf16_t* op = o + obase + (int64_t)qi[u] * a.o_st + g * (HD / 4);
#pragma unroll
        for (int d = 0; d < HD / 4; d += 4) {
            if (g * (HD / 4) + d >= a.hd) continue;
            const f32x4 x = load4(gp + d), y = load4(op + d);
            t += x[0] * y[0] + x[1] * y[1] + x[2] * y[2] + x[3] * y[3];
        }
        dl[u] = group_sum(t);
        li_lse[u] = lse[(int64_t)bh * a.T + qi[u]] * 1.4426950408889634f;      // base-2 domain: p = exp2(s c2 - lse2), one fma + v_exp_f32
        if (g == 0) delta_out[(int64_t)bh * a.T + qi[u]] = dl[u];
    }
    bf16x8 qf[G][KS], gf[G][KS];
    f32x4 acc[G][DT];
#pragma unroll
    for (int u = 0; u < G; ++u)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) acc[u][dt] = f32x4{0, 0, 0, 0};
    BlkRegs<HD> kr, vr;                                           // the NEXT key block, in flight in registers
    blk_prefetch<HD>(kr, k + base, a.q_st, a.hd);
    blk_prefetch<HD>(vr, v + base, a.q_st, a.hd);
    for (int kb = 0; kb < a.T; kb += 64) {
        __syncthreads();
        blk_commit<HD>(kr, kimg);
        blk_commit<HD>(vr, vimg);
        DMA_WAIT_SYNC();                                     // (the Q / dO images' LDS-DMA, first block; then a plain barrier)
        if (kb + 64 < a.T) {
            blk_prefetch<HD>(kr, k + base + (int64_t)(kb + 64) * a.q_st, a.q_st, a.hd);
            blk_prefetch<HD>(vr, v + base + (int64_t)(kb + 64) * a.q_st, a.q_st, a.hd);
        }
        if (kb == 0) {
#pragma unroll
            for (int u = 0; u < G; ++u)
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    qf[u][s] = frag_rows<HD>(qimg + u * IMG, 16 * wid, s, lane);
                    gf[u][s] = frag_rows<HD>(gimg + u * IMG, 16 * wid, s, lane);
                }
        }
        f32x4 ds[G][4];
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            f32x4 c[G], d[G];
#pragma unroll
            for (int u = 0; u < G; ++u) c[u] = d[u] = f32x4{0, 0, 0, 0};
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const bf16x8 kfr = frag_rows<HD>(kimg, 16 * jt, s, lane), vfr = frag_rows<HD>(vimg, 16 * jt, s, lane);
#pragma unroll
                for (int u = 0; u < G; ++u) {
                    c[u] = MFMA(kfr, qf[u][s], c[u]);
                    d[u] = MFMA(vfr, gf[u][s], d[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < G; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) ds[u][jt][r] = a.scale * __builtin_amdgcn_exp2f(c[u][r] * c2 - li_lse[u]) * (d[u][r] - dl[u]);
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            bf16x8 sf[G];
#pragma unroll
            for (int u = 0; u < G; ++u) sf[u] = pack_acc(ds[u][2 * s2], ds[u][2 * s2 + 1]);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const bf16x8 kc = frag_cols_perm<HD>(kimg, 16 * dt, 32 * s2, lane);
#pragma unroll
                for (int u = 0; u < G; ++u) acc[u][dt] = MFMA(kc, sf[u], acc[u][dt]);
            }
        }
    }
    __syncthreads();                                              // the Q / dO images are dead (their fragments sit in registers): dq over Q
#pragma unroll
    for (int u = 0; u < G; ++u) out_stage16<HD, DT>(qimg, 64 * u + 16 * wid, acc[u], lane);
    __syncthreads();
    out_flush<HD>(qimg, 64 * G, dq + base + (int64_t)qb * a.q_st, a.q_st, a.hd);
    if (cs_part) {     // per-(sample, query block) column sums of dq -> cs_part[b * gridDim.x + block][0 * H hd + h hd + c]
        float* cs = reinterpret_cast<float*>(gimg);                   // (the dO images are dead too; the staged dq tile sits in the Q images)
        f32x4 t[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {         // groups folded in order, each value rounded as stored
            t[dt] = f32x4{0, 0, 0, 0};
#pragma unroll
            for (int u = 0; u < G; ++u) t[dt] += as_stored_bf16(acc[u][dt]);
        }
        attn_cs_wave<DT>(t, cs + wid * HD, lane);
        __syncthreads();
        attn_cs_commit<HD>(cs, 1, 0, a.hd, a.H * a.hd, h, cs_part + ((int64_t)b * gridDim.x + blockIdx.x) * cs_ld);
    }
}

// ------------------------------------------------------------------------------------------------
// backward, key-major: grid (T/(64 G), B*H); dK and dV for 64 G keys, queries stream (G as above)
// ------------------------------------------------------------------------------------------------
template <int HD, int G>
__global__ void __launch_bounds__(256)
attn_bwd_dkv_mfma(AttnMfmaArgs a, const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                  const bf16_t* __restrict__ d_o, const float* __restrict__ lse, const float* __restrict__ delta,
                  bf16_t* __restrict__ dk, bf16_t* __restrict__ dv, float* __restrict__ cs_part = nullptr, int64_t cs_ld = 0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 G + 2 images of 64 x HD bf16 + lse / delta of the query block
    constexpr int KS = HD / 32, DT = HD / 16, IMG = Img<HD>::BYTES;
    char* kimg = smem;                       // G images
    char* vimg = kimg + G * IMG;             // G images
    char* qimg = vimg + G * IMG;
    char* gimg = qimg + IMG;
    float* lse_s = reinterpret_cast<float*>(gimg + IMG);
    float* del_s = lse_s + 64;
    const float c2 = a.scale * 1.4426950408889634f;
    const int lane = threadIdx.x & 63, g = lane >> 4;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int bh = blockIdx.y, b = bh / a.H, h = bh % a.H;
    const int jb = blockIdx.x * 64 * G;
    const int64_t base = b * a.q_sb + h * a.q_sh, obase = b * a.o_sb + h * a.o_sh;
#pragma unroll
    for (int u = 0; u < G; ++u) {
        stage_block<HD>(k + base + (int64_t)(jb + 64 * u) * a.q_st, a.q_st, kimg + u * IMG, wid, lane, a.hd);
        stage_block<HD>(v + base + (int64_t)(jb + 64 * u) * a.q_st, a.q_st, vimg + u * IMG, wid, lane, a.hd);
    }
    bf16x8 kf[G][KS], vf[G][KS];
    f32x4 av[G][DT], ak[G][DT];
#pragma unroll
    for (int u = 0; u < G; ++u)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) av[u][dt] = ak[u][dt] = f32x4{0, 0, 0, 0};
    BlkRegs<HD> qr, gr;                                           // the NEXT query block (Q, dO, lse, delta), in flight in registers
    float lse_r = 0.f, del_r = 0.f;
    blk_prefetch<HD>(qr, q + base, a.q_st, a.hd);
    blk_prefetch<HD>(gr, d_o + obase, a.o_st, a.hd);
    if (threadIdx.x < 64) {
        lse_r = lse[(int64_t)bh * a.T + threadIdx.x];
        del_r = delta[(int64_t)bh * a.T + threadIdx.x];
    }
    for (int ib = 0; ib < a.T; ib += 64) {
        __syncthreads();
        blk_commit<HD>(qr, qimg);
        blk_commit<HD>(gr, gimg);
        if (threadIdx.x < 64) {
            lse_s[threadIdx.x] = lse_r * 1.4426950408889634f;                 // base-2 domain
            del_s[threadIdx.x] = del_r;
        }
        DMA_WAIT_SYNC();                                     // (the K / V images' LDS-DMA, first block; then a plain barrier)
        if (ib + 64 < a.T) {
            blk_prefetch<HD>(qr, q + base + (int64_t)(ib + 64) * a.q_st, a.q_st, a.hd);
            blk_prefetch<HD>(gr, d_o + obase + (int64_t)(ib + 64) * a.o_st, a.o_st, a.hd);
            if (threadIdx.x < 64) {
                lse_r = lse[(int64_t)bh * a.T + ib + 64 + threadIdx.x];
                del_r = delta[(int64_t)bh * a.T + ib + 64 + threadIdx.x];
            }
        }
        if (ib == 0) {
#pragma unroll
            for (int u = 0; u < G; ++u)
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    kf[u][s] = frag_rows<HD>(kimg + u * IMG, 16 * wid, s, lane);
                    vf[u][s] = frag_rows<HD>(vimg + u * IMG, 16 * wid, s, lane);
                }
        }
        f32x4 p[G][4], ds[G][4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            f32x4 c[G], d[G];
#pragma unroll
            for (int u = 0; u < G; ++u) c[u] = d[u] = f32x4{0, 0, 0, 0};
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const bf16x8 qfr = frag_rows<HD>(qimg, 16 * it, s, lane), gfr = frag_rows<HD>(gimg, 16 * it, s, lane);
#pragma unroll
                for (int u = 0; u < G; ++u) {
                    c[u] = MFMA(qfr, kf[u][s], c[u]);
                    d[u] = MFMA(gfr, vf[u][s], d[u]);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * it + 4 * g + r;
                const float ls = lse_s[i], de = del_s[i];
#pragma unroll
                for (int u = 0; u < G; ++u) {
                    const float pr = __builtin_amdgcn_exp2f(c[u][r] * c2 - ls);
                    p[u][it][r] = pr;
                    ds[u][it][r] = a.scale * pr * (d[u][r] - de);
                }
            }
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            bf16x8 pf[G], sf[G];
#pragma unroll
            for (int u = 0; u < G; ++u) {
                pf[u] = pack_acc(p[u][2 * s2], p[u][2 * s2 + 1]);
                sf[u] = pack_acc(ds[u][2 * s2], ds[u][2 * s2 + 1]);
            }
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const bf16x8 gc = frag_cols_perm<HD>(gimg, 16 * dt, 32 * s2, lane), qc = frag_cols_perm<HD>(qimg, 16 * dt, 32 * s2, lane);
#pragma unroll
                for (int u = 0; u < G; ++u) {
                    av[u][dt] = MFMA(gc, pf[u], av[u][dt]);
                    ak[u][dt] = MFMA(qc, sf[u], ak[u][dt]);
                }
            }
        }
    }
    __syncthreads();                                              // the K / V images are dead (fragments in registers): dk over K, dv over V
#pragma unroll
    for (int u = 0; u < G; ++u) {
        out_stage16<HD, DT>(kimg, 64 * u + 16 * wid, ak[u], lane);
        out_stage16<HD, DT>(vimg, 64 * u + 16 * wid, av[u], lane);
    }
    __syncthreads();
    out_flush<HD>(kimg, 64 * G, dk + base + (int64_t)jb * a.q_st, a.q_st, a.hd);
    out_flush<HD>(vimg, 64 * G, dv + base + (int64_t)jb * a.q_st, a.q_st, a.hd);
    if (cs_part) {     // per-(sample, key block) column sums of dk | dv -> columns H hd + h hd + c and 2 H hd + h hd + c
        float* cs = reinterpret_cast<float*>(qimg);                    // [2][4 waves][HD] floats <= 2 images (Q / dO: dead after the last block's barrier above)
        f32x4 tk[DT], tv[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            tk[dt] = tv[dt] = f32x4{0, 0, 0, 0};
#pragma unroll
            for (int u = 0; u < G; ++u) {
                tk[dt] += as_stored_bf16(ak[u][dt]);
                tv[dt] += as_stored_bf16(av[u][dt]);
            }
        }
        attn_cs_wave<DT>(tk, cs + wid * HD, lane);
        attn_cs_wave<DT>(tv, cs + 4 * HD + wid * HD, lane);
        __syncthreads();
        attn_cs_commit<HD>(cs, 2, 1, a.hd, a.H * a.hd, h, cs_part + ((int64_t)b * gridDim.x + blockIdx.x) * cs_ld);
    }
}

// ------------------------------------------------------------------------------------------------
// backward for T == 64 (DiT-B/4's 8x8 patch grid): one workgroup per (sample, head) holds all of Q, K, V, dO, so
// delta, dQ, dK and dV come from ONE launch and one set of loads (the two kernels above each re-stage everything and
// are latency-bound at this size).  Phase 1 = attn_bwd_dq_mfma's body, phase 2 = attn_bwd_dkv_mfma's, lse / delta
// handed over through LDS.
// ------------------------------------------------------------------------------------------------
#ifndef ATTN64_PROF
#define ATTN64_PROF 0
#endif
#if ATTN64_PROF
__device__ unsigned long long attn64_prof_buf[64];
extern "C" int vaw_debug_attn64_prof(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(attn64_prof_buf), sizeof(unsigned long long) * n);
}
#define A64_STAMP(slot)                                                                                               \
    do {                                                                                                              \
        if (threadIdx.x == 0 && blockIdx.x == 1500 && prof_n < 60)                                                    \
            attn64_prof_buf[prof_n++] = ((unsigned long long)(slot) << 56) | (wall_clock64() & 0xffffffffffffffull);  \
    } while (0)
#else
#define A64_STAMP(slot) do {} while (0)
#endif
template <int HD>
__global__ void __launch_bounds__(256)
attn_bwd_t64_mfma(AttnMfmaArgs a, const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                  const bf16_t* __restrict__ o, const bf16_t* __restrict__ d_o, const float* __restrict__ lse,
                  float* __restrict__ delta_out, bf16_t* __restrict__ dq, bf16_t* __restrict__ dk, bf16_t* __restrict__ dv,
                  float* __restrict__ cs_part = nullptr, int64_t cs_ld = 0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // 4 images of 64 x HD bf16 (reused as output staging; lse / delta: below)
    constexpr int KS = HD / 32, DT = HD / 16;
#if ATTN64_PROF
    int prof_n = 0;
#endif
    A64_STAMP(1);
    char* qimg = smem;
    char* gimg = qimg + Img<HD>::BYTES;
    char* kimg = gimg + Img<HD>::BYTES;
    char* vimg = kimg + Img<HD>::BYTES;
    // (r4) the rows' lse / delta, which phase 2 reads for ALL 64 queries, live in the V image once phase 1 is through with it (they wait in
    // registers until then) instead of in 512 bytes of their own: 4 x 8 KiB = 32 KiB exactly at hd 64 -> FIVE workgroups per CU, not four
    float* lse_s = reinterpret_cast<float*>(vimg);
    float* del_s = lse_s + 64;
    const int lane = threadIdx.x & 63, g = lane >> 4, li = lane & 15;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // (ATTN64_XCD=1: workgroups are dealt round-robin over the 8 XCDs; give each XCD a CONTIGUOUS run of (sample, head) pairs, so that
    // the 128-byte row pieces of a sample's heads -- adjacent in memory -- are asked for by one L2 at about the same time)
    int bh = blockIdx.x;
    if (ATTN64_XCD && (gridDim.x & 7) == 0) bh = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int b = bh / a.H, h = bh % a.H;
    const int64_t base = b * a.q_sb + h * a.q_sh, obase = b * a.o_sb + h * a.o_sh;
    stage_block<HD>(q + base, a.q_st, qimg, wid, lane, a.hd);
    stage_block<HD>(d_o + obase, a.o_st, gimg, wid, lane, a.hd);
    stage_block<HD>(k + base, a.q_st, kimg, wid, lane, a.hd);
    stage_block<HD>(v + base, a.q_st, vimg, wid, lane, a.hd);
    const int qi = 16 * wid + li;
    // delta = rowsum(dO * O) of this lane's query: O comes from global (requested now, 16 bytes per lane and k-step at the columns of
    // this lane's dO fragment), dO from its LDS image once that has landed -- the kernel used to read dO twice (image + registers)
    bf16x8 of[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int c = 32 * s + 8 * g;
        const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        of[s] = c < a.hd ? *reinterpret_cast<const bf16x8*>(o + obase + (int64_t)qi * a.o_st + c) : z;
    }
    float dl = 0.f;
    const float c2 = a.scale * 1.4426950408889634f;
    const float li_lse = lse[(int64_t)bh * 64 + qi] * 1.4426950408889634f;             // base-2 domain
    A64_STAMP(2);
    DMA_WAIT_SYNC();
    A64_STAMP(3);
    // Output staging (r3): dq, dk, dv leave through LDS as whole rows -- a lane of an accumulator tile owns 4 bf16 of one row, so
    // direct stores write 32-byte pieces of 16 rows per instruction; staged, eight lanes write one 128-byte row with 16 bytes each
    // -- and their column sums (the qkv bias gradient) are taken from the staged tiles by 3 hd threads in a fixed row order
    // instead of by 192 shuffles per wave.  Tile layout: [64 rows][2 HD bytes], 16-byte chunk c of row r at c ^ (r & 7) (HD 96:
    // the 208-byte pitch of the images, no XOR); dq is staged over the K (and V) image once every wave holds its phase-2
    // fragments, dk / dv over the Q / dO images after phase 2.
    auto st_off = [](int r, int c16) { return HD == 96 ? r * 208 + 16 * c16 : r * (2 * HD) + ((c16 ^ (r & (HD / 8 < 8 ? HD / 8 - 1 : 7))) << 4); };
    auto stage_tile = [&](char* dst, const f32x4 (&acc)[DT]) {     // this wave's 16 rows: lane (li, g) -> row 16 wid + li, columns 16 dt + 4 g ..
        const int r = 16 * wid + li;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const bf16x4 w = {(bf16_t)acc[dt][0], (bf16_t)acc[dt][1], (bf16_t)acc[dt][2], (bf16_t)acc[dt][3]};
            *reinterpret_cast<bf16x4*>(dst + st_off(r, 2 * dt + (g >> 1)) + 8 * (g & 1)) = w;
        }
    };
    f32x4 acc[DT];
    // ---- phase 1: dQ of this wave's 16 queries ----
    {
        bf16x8 qf[KS], gf[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            qf[s] = frag_rows<HD>(qimg, 16 * wid, s, lane);
            gf[s] = frag_rows<HD>(gimg, 16 * wid, s, lane);
        }
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int e = 0; e < 8; ++e) dl += (float)gf[s][e] * (float)of[s][e];      // (columns beyond hd are zero in the image)
        dl = group_sum(dl);
        if (g == 0) delta_out[(int64_t)bh * 64 + qi] = dl;
        f32x4 ds[4];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) acc[dt] = f32x4{0, 0, 0, 0};
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            f32x4 c = {0, 0, 0, 0}, d = {0, 0, 0, 0};
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                c = MFMA(frag_rows<HD>(kimg, 16 * jt, s, lane), qf[s], c);
                d = MFMA(frag_rows<HD>(vimg, 16 * jt, s, lane), gf[s], d);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) ds[jt][r] = a.scale * __builtin_amdgcn_exp2f(c[r] * c2 - li_lse) * (d[r] - dl);
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const bf16x8 sf = pack_acc(ds[2 * s2], ds[2 * s2 + 1]);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) acc[dt] = MFMA(frag_cols_perm<HD>(kimg, 16 * dt, 32 * s2, lane), sf, acc[dt]);
        }
    }
    A64_STAMP(4);
    // ---- phase 2: dK, dV of this wave's 16 keys (lse_s / del_s were written before the sync) ----
    f32x4 av[DT], ak[DT];
    {
        bf16x8 kf[KS], vf[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            kf[s] = frag_rows<HD>(kimg, 16 * wid, s, lane);
            vf[s] = frag_rows<HD>(vimg, 16 * wid, s, lane);
        }
        __syncthreads();                  // nobody reads the K / V images any more: dq is staged over K, lse / delta go into V
        stage_tile(kimg, acc);
        if (g == 0) {
            lse_s[qi] = li_lse;
            del_s[qi] = dl;
        }
        __syncthreads();
        f32x4 p[4], ds[4];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) av[dt] = ak[dt] = f32x4{0, 0, 0, 0};
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            f32x4 c = {0, 0, 0, 0}, d = {0, 0, 0, 0};
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                c = MFMA(frag_rows<HD>(qimg, 16 * it, s, lane), kf[s], c);
                d = MFMA(frag_rows<HD>(gimg, 16 * it, s, lane), vf[s], d);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * it + 4 * g + r;
                const float pr = __builtin_amdgcn_exp2f(c[r] * c2 - lse_s[i]);
                p[it][r] = pr;
                ds[it][r] = a.scale * pr * (d[r] - del_s[i]);
            }
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const bf16x8 pf = pack_acc(p[2 * s2], p[2 * s2 + 1]);
            const bf16x8 sf = pack_acc(ds[2 * s2], ds[2 * s2 + 1]);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                av[dt] = MFMA(frag_cols_perm<HD>(gimg, 16 * dt, 32 * s2, lane), pf, av[dt]);
                ak[dt] = MFMA(frag_cols_perm<HD>(qimg, 16 * dt, 32 * s2, lane), sf, ak[dt]);
            }
        }
    }
    A64_STAMP(5);
    __syncthreads();                      // nobody reads the Q / dO images any more
    stage_tile(qimg, ak);
    stage_tile(gimg, av);
    __syncthreads();
    A64_STAMP(6);
    // whole rows out: tile 0 = dq (staged at kimg), 1 = dk (qimg), 2 = dv (gimg); 16 bytes per thread and access
    constexpr int CPR = HD / 8;
#pragma unroll
    for (int i = 0; i < (3 * 64 * CPR + 255) / 256; ++i) {
        const int idx = threadIdx.x + 256 * i;
        if ((3 * 64 * CPR) % 256 != 0 && idx >= 3 * 64 * CPR) break;
        const int tile = idx / (64 * CPR), rem = idx - tile * (64 * CPR), r = rem / CPR, c16 = rem - r * CPR;
        if (8 * c16 >= a.hd) continue;
        const char* src = (tile == 0 ? kimg : tile == 1 ? qimg : gimg) + st_off(r, c16);
        bf16_t* dst = (tile == 0 ? dq : tile == 1 ? dk : dv) + base + (int64_t)r * a.q_st + 8 * c16;
        *reinterpret_cast<bf16x8*>(dst) = *reinterpret_cast<const bf16x8*>(src);
    }
    A64_STAMP(7);
    if (cs_part) {     // this (sample, head)'s column sums of dq | dk | dv (as stored, rows in ascending order) -> its columns of the
                       // sample's partial row [3 H hd]
        for (int t = threadIdx.x; t < 3 * HD; t += 256) {
            const int tile = t / HD, c = t - tile * HD;
            if (c >= a.hd) continue;
            const char* src = (tile == 0 ? kimg : tile == 1 ? qimg : gimg) + 2 * (c & 7);
            float sum = 0.f;
#pragma unroll 8
            for (int r = 0; r < 64; ++r) sum += (float)*reinterpret_cast<const bf16_t*>(src + st_off(r, c >> 3));
            cs_part[(int64_t)b * cs_ld + tile * (a.H * a.hd) + h * a.hd + c] = sum;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host dispatch (called from attention.hip)
// ------------------------------------------------------------------------------------------------
static bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

bool vaw_attn_mfma_ok(vaw_dtype dt, const vaw_attn_desc* d, const void* q, const void* k, const void* v, const void* o) {
    const bool hd_ok = d->hd >= 8 && d->hd <= 128 && d->hd % 8 == 0;      // padded up to 32 / 64 / 96 / 128 in LDS
    return dt == VAW_BF16 && hd_ok && d->T % 64 == 0 && d->q_sd == 1 && d->o_sd == 1 && d->q_st % 8 == 0 && d->q_sh % 8 == 0 &&
           d->q_sb % 8 == 0 && d->o_st % 8 == 0 && d->o_sh % 8 == 0 && d->o_sb % 8 == 0 && aligned16(q) && aligned16(k) &&
           aligned16(v) && aligned16(o) && (int64_t)d->B * d->H < 65536;
}

static AttnMfmaArgs mk_args(const vaw_attn_desc* d) {
    AttnMfmaArgs a{d->B, d->H, d->T, d->q_sb, d->q_sh, d->q_st, d->o_sb, d->o_sh, d->o_st, d->scale, d->hd};
    return a;
}

static bool attn_qg2() {
    static int on = -1;
    if (on < 0) { const char* v = getenv("VAW_ATTN_QG2"); on = v ? atoi(v) : 1; }
    return on != 0;
}

#define DISPATCH_HD(hd, ...)                                \
    switch (((hd) + 31) / 32) {                             \
        case 1: { constexpr int HD = 32; __VA_ARGS__ } break;    \
        case 2: { constexpr int HD = 64; __VA_ARGS__ } break;    \
        case 3: { constexpr int HD = 96; __VA_ARGS__ } break;    \
        default: { constexpr int HD = 128; __VA_ARGS__ } break;  \
    }

bool vaw_attn_fwd_big(const vaw_attn_desc* d, const void* q, const void* k, const void* v, void* o, float* lse, hipStream_t s);

int vaw_attn_fwd_mfma(const vaw_attn_desc* d, const void* q, const void* k, const void* v, void* o, float* lse,
                      hipStream_t s) {
    if (vaw_attn_fwd_big(d, q, k, v, o, lse, s)) {        // attention_bwd_big.hip
        VAW_CHECK_LAUNCH("attn_fwd_big");
        return VAW_OK;
    }
    AttnMfmaArgs a = mk_args(d);
    // two 16-query groups per wave: +10..17 % for head dims <= 64 (tools/attn_bench.py), +30 % for the padded 96-wide images
    // (DiT-XL's 72, UNet_64's 96) now that their rows are conflict-free; 128-wide images keep one group (accumulators)
    if (d->T % 128 == 0 && d->hd <= 96 && attn_qg2()) {
        dim3 grid(d->T / 128, d->B * d->H);
        DISPATCH_HD(d->hd,
            const int lds = (HD <= 64 ? 6 : 4) * Img<HD>::BYTES;
            (void)hipFuncSetAttribute((const void*)attn_fwd_mfma<HD, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            attn_fwd_mfma<HD, 2><<<grid, 256, lds, s>>>(a, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)o, lse);
        )
        VAW_CHECK_LAUNCH("attn_fwd_mfma");
        return VAW_OK;
    }
    dim3 grid(d->T / 64, d->B * d->H);
    if (d->T == 64) {        // a single key block: nothing to double-buffer, keep the LDS footprint (and residency) small
        DISPATCH_HD(d->hd,
            const int lds = 3 * Img<HD>::BYTES;
            (void)hipFuncSetAttribute((const void*)attn_fwd_mfma<HD, 1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            attn_fwd_mfma<HD, 1, false><<<grid, 256, lds, s>>>(a, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)o, lse);
        )
        VAW_CHECK_LAUNCH("attn_fwd_mfma");
        return VAW_OK;
    }
    DISPATCH_HD(d->hd,
        const int lds = (HD <= 64 ? 5 : 3) * Img<HD>::BYTES;
        (void)hipFuncSetAttribute((const void*)attn_fwd_mfma<HD, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attn_fwd_mfma<HD, 1><<<grid, 256, lds, s>>>(a, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)o, lse);
    )
    VAW_CHECK_LAUNCH("attn_fwd_mfma");
    return VAW_OK;
}

bool vaw_attn_bwd_big(const vaw_attn_desc* d, const void* q, const void* k, const void* v, const void* o, const void* d_o,
                      const float* lse, float* delta, void* dq, void* dk, void* dv, hipStream_t s, float* cs_part, int64_t* cs_rows_out);

// cs_part (may be NULL): [cs_rows][3 H hd] f32 partial column sums of dq | dk | dv, one row per (sample, 64 G-token block);
// *cs_rows_out receives the row count
int vaw_attn_bwd_mfma(const vaw_attn_desc* d, const void* q, const void* k, const void* v, const void* o, const void* d_o,
                      const float* lse, float* delta, void* dq, void* dk, void* dv, hipStream_t s, float* cs_part, int64_t* cs_rows_out) {
    AttnMfmaArgs a = mk_args(d);
    const int64_t cs_ld = 3LL * d->H * d->hd;
    dim3 grid(d->T / 64, d->B * d->H);
    if (d->T == 64) {      // single block of queries and keys: one fused launch
        DISPATCH_HD(d->hd,
            const int lds = 4 * Img<HD>::BYTES;
            (void)hipFuncSetAttribute((const void*)attn_bwd_t64_mfma<HD>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            attn_bwd_t64_mfma<HD><<<d->B * d->H, 256, lds, s>>>(a, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (const bf16_t*)o,
                                                              (const bf16_t*)d_o, lse, delta, (bf16_t*)dq, (bf16_t*)dk, (bf16_t*)dv, cs_part, cs_ld);
        )
        if (cs_rows_out) *cs_rows_out = d->B;
        VAW_CHECK_LAUNCH("attn_bwd_t64_mfma");
        return VAW_OK;
    }
    // T a multiple of 256, head dims 40 .. 96: the 64-rows-per-wave pair (attention_bwd_big.hip)
    if (vaw_attn_bwd_big(d, q, k, v, o, d_o, lse, delta, dq, dk, dv, s, cs_part, cs_rows_out)) {
        VAW_CHECK_LAUNCH("attn_bwd_big");
        return VAW_OK;
    }
    static int g2 = -1;
    if (g2 < 0) { const char* e = getenv("VAW_ATTN_BWD_G2"); g2 = e ? atoi(e) : 1; }
#define ATTN_BWD_GO(Gv)                                                                                                          \
    do {                                                                                                                         \
        dim3 gridg(d->T / (64 * Gv), d->B * d->H);                                                                               \
        const int lds = (2 * Gv + 2) * Img<HD>::BYTES + 2 * 64 * 4;                                                              \
        (void)hipFuncSetAttribute((const void*)attn_bwd_dq_mfma<HD, Gv>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);       \
        (void)hipFuncSetAttribute((const void*)attn_bwd_dkv_mfma<HD, Gv>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);      \
        attn_bwd_dq_mfma<HD, Gv><<<gridg, 256, lds, s>>>(a, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (const bf16_t*)o, \
                                                        (const bf16_t*)d_o, lse, delta, (bf16_t*)dq, cs_part, cs_ld);            \
        attn_bwd_dkv_mfma<HD, Gv><<<gridg, 256, lds, s>>>(a, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (const bf16_t*)d_o, \
                                                         lse, delta, (bf16_t*)dk, (bf16_t*)dv, cs_part, cs_ld);                  \
        if (cs_rows_out) *cs_rows_out = (int64_t)d->B * (d->T / (64 * Gv));                                                      \
    } while (0)
    // measured (tools/attn_bench.py, T = 256 / 1024): two groups are +10 % on the 96-wide images with 96 real channels
    // (UNet_64), neutral on DiT-XL's 72-in-96, and 10-20 % SLOWER on 64-wide images, where the third resident workgroup is
    // worth more than the shared fragments: these kernels are occupancy-, not DMA-latency-bound
    if (d->T % 128 == 0 && d->hd > 64 && d->hd <= 96 && g2) {
        DISPATCH_HD(d->hd, if constexpr (HD == 96) ATTN_BWD_GO(2); else ATTN_BWD_GO(1);)
    } else {
        DISPATCH_HD(d->hd, ATTN_BWD_GO(1);)
    }
    VAW_CHECK_LAUNCH("attn_bwd_mfma");
    return VAW_OK;
}
