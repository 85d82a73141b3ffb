// Instantiations of the persistent GEMM kernel (gemm_p8_kernel.h) for the weight-gradient layout: A [K][M] (dy), B [K][N] (x).
#include "gemm_p8_kernel.h"

void p8_launch_wgrad(const P8Launch& L, const EpiDev& e, hipStream_t s) {
    switch (L.epi) {
        P8_CASE(false, false, P8_SLAB);
        P8_CASE(false, false, P8_STORE);      // un-split plain results: the packed adaLN weight gradient (K = batch; round 3 ran it on P8_ANY: 116 us)
        default:
            if (L.ntw == 4) p8_launch_one<false, false, 4, P8_ANY>(L.a, L.lda, L.b, L.ldb, L.nk, L.tiles_m, L.tiles_n, L.split, L.grid, e, s, L.team_delay);
            else p8_launch_one<false, false, 3, P8_ANY>(L.a, L.lda, L.b, L.ldb, L.nk, L.tiles_m, L.tiles_n, L.split, L.grid, e, s, L.team_delay);
    }
}

// A [K][M], B [N][K]: no launch of the training step has this layout; kept for completeness of vaw_gemm
void p8_launch_tn(const P8Launch& L, const EpiDev& e, hipStream_t s) {
    switch (L.epi) {
        P8_CASE(false, true, P8_SLAB);
        default:
            if (L.ntw == 4) p8_launch_one<false, true, 4, P8_ANY>(L.a, L.lda, L.b, L.ldb, L.nk, L.tiles_m, L.tiles_n, L.split, L.grid, e, s, L.team_delay);
            else p8_launch_one<false, true, 3, P8_ANY>(L.a, L.lda, L.b, L.ldb, L.nk, L.tiles_m, L.tiles_n, L.split, L.grid, e, s, L.team_delay);
    }
}
