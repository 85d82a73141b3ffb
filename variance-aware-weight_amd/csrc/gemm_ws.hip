// Warp-specialised persistent GEMM (gemm_ws_kernel.h): dispatch and the forward-layout instantiations; the input-gradient layout
// lives in gemm_ws_dgrad.hip.
#include "gemm_ws_kernel.h"

// VAW_WS_LOADERS=8: the 192-column kernels with eight loader waves (16 waves per workgroup) instead of four
static bool ws_loaders8() {
    static int v = -1;
    if (v < 0) { const char* s = getenv("VAW_WS_LOADERS"); v = (s && atoi(s) == 8) ? 1 : 0; }
    return v == 1;
}

void ws_launch_dgrad(int ntw, int epi, const bf16_t* a, int64_t lda, const bf16_t* b, int64_t ldb, int nk, int tiles_m, int tiles_n,
                     int grid, const EpiDev& e, hipStream_t s);

void vaw_ws_launch(int ntw, int epi, int b_kmajor, int64_t M, int64_t N, int64_t K, const bf16_t* a, int64_t lda, const bf16_t* b,
                   int64_t ldb, const EpiDev& e, int cus, hipStream_t s) {
    const int bn = 64 * ntw;
    const int tiles_m = (int)((M + WS_BM - 1) / WS_BM), tiles_n = (int)((N + bn - 1) / bn), nk = (int)(K / 64);
    const int64_t items = (int64_t)tiles_m * tiles_n;
    const int grid = (int)(items < cus ? items : cus);
    if (!b_kmajor) { ws_launch_dgrad(ntw, epi, a, lda, b, ldb, nk, tiles_m, tiles_n, grid, e, s); return; }
#define WS_CASE(EPIv)                                                                                   \
    case EPIv:                                                                                          \
        if (ntw == 4) ws_launch_one<true, 4, EPIv>(a, lda, b, ldb, nk, tiles_m, tiles_n, grid, e, s);   \
        else if (ws_loaders8()) ws_launch_one<true, 3, EPIv, 8>(a, lda, b, ldb, nk, tiles_m, tiles_n, grid, e, s);   \
        else ws_launch_one<true, 3, EPIv>(a, lda, b, ldb, nk, tiles_m, tiles_n, grid, e, s);            \
        break
    switch (epi) {
        WS_CASE(P8_STORE);
        WS_CASE(P8_GELU);
        WS_CASE(P8_GATE);
        default: break;
    }
}
