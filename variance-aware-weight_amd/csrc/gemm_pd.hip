// Parked-drain persistent GEMM (gemm_pd_kernel.h): shape planning, dispatch and the forward-layout instantiations
// (A [M][K] activations, B [N][K] weights).  The input-gradient layout lives in gemm_pd_dgrad.hip.
#include "gemm_pd_kernel.h"

void pd_launch_dgrad(int ntw, int epi, const bf16_t* a, int64_t lda, const bf16_t* b, int64_t ldb, int nk, int tiles_m, int tiles_n,
                     int grid, const EpiDev& e, hipStream_t s);

static int pd_num_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

// Epilogue kind this kernel offers for the launch, or -1.  (Same classification as vaw_p8_launch; K-split launches, f32 plain
// outputs, fused row sums and the UNet's residual kinds stay with gemm_p8_kernel.)
int vaw_pd_epi_kind(const EpiDev& e, bool a_kmajor, bool b_kmajor, int64_t M, int64_t N, int64_t K) {
    // (K >= 12 K tiles: the longest drain -- 8 steps + 2 slots of operand lead -- and the two K tiles behind it are unrolled in front of the K loop)
    if (!a_kmajor || K % 64 != 0 || K / 64 < 12 || N % 8 != 0 || M < PD_BM) return -1;
    const bool bf16_out = !e.out_f32;
    if (e.act == 1 && e.aux_out && !e.gate && !e.resid && !e.rowadd && bf16_out && !e.colpart && b_kmajor) return P8_GELU;
    if (e.act == 2 && !e.bias && !e.aux_out && !e.gate && !e.resid && !e.rowadd && bf16_out && e.alpha == 1.f && !b_kmajor) return P8_DGELU;
    if (e.act == 0 && e.gate && e.resid && !e.resid_act && e.aux_out && !e.rowadd && e.out_f32 && e.beta == 0.f && !e.colpart && b_kmajor &&
        e.rpb % 8 == 0)
        return P8_GATE;
    if (e.act == 0 && !e.aux_out && !e.gate && !e.resid && !e.rowadd && e.beta == 0.f && bf16_out) return P8_STORE;
    return -1;
}

// tile width: the one with fewer rounds of workgroups; a 192-column item costs ~0.8 of a 256-column one
int vaw_pd_pick_ntw(int64_t M, int64_t N, int cus_avail) {
    const int cus = cus_avail > 0 ? cus_avail : pd_num_cus();
    double best = 1e30;
    int ntw = 4;
    for (int t = 4; t >= 3; --t) {
        const int bn = 64 * t;
        if (t == 3 && ((N + 191) / 192) * 192 > ((N + 255) / 256) * 256) continue;
        const int64_t items = ((M + PD_BM - 1) / PD_BM) * ((N + bn - 1) / bn);
        const double c = (double)((items + cus - 1) / cus) * (t == 4 ? 1.0 : 0.8);
        if (c < best - 1e-9) { best = c; ntw = t; }
    }
    return ntw;
}

void vaw_pd_launch(int ntw, int epi, int b_kmajor, int64_t M, int64_t N, int64_t K, const bf16_t* a, int64_t lda, const bf16_t* b,
                   int64_t ldb, const EpiDev& e, int cus_avail, hipStream_t s) {
    const int bn = 64 * ntw;
    const int tiles_m = (int)((M + PD_BM - 1) / PD_BM), tiles_n = (int)((N + bn - 1) / bn), nk = (int)(K / 64);
    const int cus = cus_avail > 0 ? cus_avail : pd_num_cus();
    const int64_t items = (int64_t)tiles_m * tiles_n;
    const int grid = (int)(items < cus ? items : cus);
    if (!b_kmajor) { pd_launch_dgrad(ntw, epi, a, lda, b, ldb, nk, tiles_m, tiles_n, grid, e, s); return; }
#define PD_CASE(EPIv)                                                                                   \
    case EPIv:                                                                                          \
        if (ntw == 4) pd_launch_one<true, 4, EPIv>(a, lda, b, ldb, nk, tiles_m, tiles_n, grid, e, s);   \
        else pd_launch_one<true, 3, EPIv>(a, lda, b, ldb, nk, tiles_m, tiles_n, grid, e, s);            \
        break
    switch (epi) {
        PD_CASE(P8_STORE);
        PD_CASE(P8_GELU);
        PD_CASE(P8_GATE);
        default: break;
    }
}
