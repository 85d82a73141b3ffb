// Implicit-GEMM conv3x3 on the persistent kernel (gemm_p8_kernel.h, CONV modes): instantiations and the host entry that
// vaw_conv3x3 (gemm.hip) tries first.  Reference: the 3x3 convolutions of ResBlock / stem / head, models/unet.py:182-213,492,625.
#include "gemm_p8_kernel.h"

struct P8Plan {
    bool use;
    int ntw, split, grid;
};
P8Plan vaw_p8_plan(int64_t M, int64_t N, int64_t K, bool plain_f32, bool want_colsum, int64_t ws_floats, int force);

// out[n][m] = beta * out[n][m] + sum_s slab[s][m][n]: the slab reduce of the transposed weight-gradient problem
// (slab rows = (tap, ci), columns = co; out = dW [Co][9*Ci]).  32 (m) x 64 (n) tiles: 16-byte slab reads (16 lanes per 256-byte row
// piece), the sums transposed through LDS, 16-byte writes of four consecutive m (8 lanes per 128-byte output row piece).  Nt % 4 == 0,
// Mt % 4 == 0 (9 Ci with Ci % 8 == 0).  (r3: the 4-byte version read 50 MB of slabs at 2.2 TB/s, 23 us per conv launch.)
__global__ void __launch_bounds__(256)
p8_conv_wgrad_reduce_kernel(const float* __restrict__ slab, int S, int Mt, int Nt, float* __restrict__ out, float beta,
                            const float* __restrict__ rowpart, float* __restrict__ bias_out, float bias_beta) {
    __shared__ float tile[32][65];
    if (rowpart && blockIdx.x == 0 && blockIdx.y == 0) {      // the bias gradient: fold the per-split column sums of dy, fixed order
        for (int n = threadIdx.x; n < Nt; n += blockDim.x) {
            float t = 0.f;
            for (int sp = 0; sp < S; ++sp) t += rowpart[(int64_t)sp * Nt + n];
            bias_out[n] = (bias_beta != 0.f ? bias_beta * bias_out[n] : 0.f) + t;
        }
    }
    const int m0 = blockIdx.x * 32, n0 = blockIdx.y * 64;
    {
        const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;          // 16 lanes x 4 columns, 16 rows per pass
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int r = pass * 16 + ty, m = m0 + r, n = n0 + 4 * tx;
            f32x4 acc = {0, 0, 0, 0};
            if (m < Mt && n < Nt) {
                const float* p = slab + (int64_t)m * Nt + n;
                const int64_t plane = (int64_t)Mt * Nt;
                for (int sp = 0; sp < S; ++sp) acc += load4(p + sp * plane);
            }
            tile[r][4 * tx] = acc[0]; tile[r][4 * tx + 1] = acc[1]; tile[r][4 * tx + 2] = acc[2]; tile[r][4 * tx + 3] = acc[3];
        }
    }
    __syncthreads();
    {
        const int mq = threadIdx.x & 7, nr = threadIdx.x >> 3;           // 8 lanes x 4 consecutive m, 32 output rows (n) per pass
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int nl = pass * 32 + nr, n = n0 + nl, m = m0 + 4 * mq;
            if (n < Nt && m < Mt) {
                float* o = out + (int64_t)n * Mt + m;
                f32x4 v = {tile[4 * mq][nl], tile[4 * mq + 1][nl], tile[4 * mq + 2][nl], tile[4 * mq + 3][nl]};
                if (beta != 0.f) v += beta * load4(o);
                store4(o, v);
            }
        }
    }
}

// mode 0 forward, 1 input gradient, 2 weight gradient.  Returns false when the shape should stay on gemm.hip's kernel.
bool vaw_p8_conv(int mode, const bf16_t* act, const bf16_t* act2, const bf16_t* w, void* out, int B, int H, int W, int Ci, int Co,
                 EpiDev e, float* workspace, int64_t workspace_floats, int force, hipStream_t s, float* bias_grad, float bias_beta,
                 int* bias_done) {
    const int64_t Mpix = (int64_t)B * H * W;
    if (Mpix * (Ci > Co ? Ci : Co) >= (1LL << 31) || Mpix % 64) return false;
    int64_t M, N, K;
    if (mode == 0) { M = Mpix; N = Co; K = 9LL * Ci; if (Ci % 64) return false; }
    else if (mode == 1) { M = Mpix; N = Ci; K = 9LL * Co; if (Co % 64) return false; }
    else { M = 9LL * Ci; N = Co; K = Mpix; if (Ci % 8 || Co % 8 || !workspace) return false; }     // transposed problem
    if (N % 8 || N < 64) return false;
    if (mode != 2 && (e.act || e.aux_out || e.gate || e.rowadd || e.out_f32 || e.beta != 0.f || e.colpart)) return false;
    if (mode != 2 && e.resid && !e.resid_act) return false;
    const P8Plan pl = vaw_p8_plan(M, N, K, mode == 2, false, workspace_floats, force);
    if (!pl.use) return false;
    if (mode == 2 && pl.split < 2) return false;                    // (always K-split in practice: K = pixels, few tiles)
    const int bn = 64 * pl.ntw, tiles_m = (int)((M + 255) / 256), tiles_n = (int)((N + bn - 1) / bn), nk = (int)(K / 64);
    const P8Conv cg{H, W, Ci, Co};
    e.M = M; e.N = N; e.ldc = N; e.C = out; e.slab = workspace; e.colpart = nullptr; e.rowpart = nullptr; e.nt_off = 1;
    *bias_done = 0;
    if (mode == 2 && bias_grad && pl.ntw == 3 && workspace_floats >= (int64_t)pl.split * M * N + (int64_t)pl.split * N) {
        // bias gradient on the same launch (192-column kernel only: the 256-column one has no registers left for it)
        e.rowpart = workspace + (int64_t)pl.split * M * N;
        *bias_done = 1;
    }
#define P8C(AKv, BKv, EPIv, CV, a, lda, b, ldb)                                                                                   \
    do {                                                                                                                          \
        if (pl.ntw == 4) p8_launch_conv<AKv, BKv, 4, EPIv, CV>(a, lda, b, ldb, nk, tiles_m, tiles_n, pl.split, pl.grid, e, cg, s); \
        else p8_launch_conv<AKv, BKv, 3, EPIv, CV>(a, lda, b, ldb, nk, tiles_m, tiles_n, pl.split, pl.grid, e, cg, s);             \
    } while (0)
    if (mode == 0) {
        if (e.resid) P8C(true, true, P8_RESID, 1, act, (int64_t)Ci, w, 9LL * Ci);
        else P8C(true, true, P8_STORE, 1, act, (int64_t)Ci, w, 9LL * Ci);
    } else if (mode == 1) {
        P8C(true, false, P8_STORE, 2, act, (int64_t)Co, w, 9LL * Ci);
    } else {
        const float beta = e.beta;
        P8C(false, false, P8_SLAB, 3, act2, (int64_t)Ci, act, (int64_t)Co);      // A = x (gathered), B = dy
        dim3 grid((unsigned)((M + 31) / 32), (unsigned)((N + 63) / 64));
        p8_conv_wgrad_reduce_kernel<<<grid, 256, 0, s>>>(workspace, pl.split, (int)M, (int)N, (float*)out, beta, e.rowpart, bias_grad, bias_beta);
    }
    return true;
}
