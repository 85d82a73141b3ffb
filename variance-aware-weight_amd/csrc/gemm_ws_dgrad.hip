// Instantiations of the warp-specialised GEMM (gemm_ws_kernel.h) for the input-gradient layout: A [M][K] (dy), B [K][N] (weights as stored).
#include "gemm_ws_kernel.h"

// VAW_WS_LOADERS=8: the 192-column kernels with eight loader waves (16 waves per workgroup) instead of four
static bool ws_loaders8() {
    static int v = -1;
    if (v < 0) { const char* s = getenv("VAW_WS_LOADERS"); v = (s && atoi(s) == 8) ? 1 : 0; }
    return v == 1;
}

void ws_launch_dgrad(int ntw, int epi, const bf16_t* a, int64_t lda, const bf16_t* b, int64_t ldb, int nk, int tiles_m, int tiles_n,
                     int grid, const EpiDev& e, hipStream_t s) {
#define WS_CASE(EPIv)                                                                                    \
    case EPIv:                                                                                           \
        if (ntw == 4) ws_launch_one<false, 4, EPIv>(a, lda, b, ldb, nk, tiles_m, tiles_n, grid, e, s);   \
        else if (ws_loaders8()) ws_launch_one<false, 3, EPIv, 8>(a, lda, b, ldb, nk, tiles_m, tiles_n, grid, e, s);   \
        else ws_launch_one<false, 3, EPIv>(a, lda, b, ldb, nk, tiles_m, tiles_n, grid, e, s);            \
        break
    switch (epi) {
        WS_CASE(P8_STORE);
        WS_CASE(P8_DGELU);
        default: break;
    }
}
