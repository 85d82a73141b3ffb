// Instantiations of the persistent GEMM kernel (gemm_p8_kernel.h) for the forward layout: A [M][K] (activations), B [N][K] (weights).
#include "gemm_p8_kernel.h"

void p8_launch_fwd(const P8Launch& L, const EpiDev& e, hipStream_t s) {
    switch (L.epi) {
        P8_CASE(true, true, P8_STORE);
        P8_CASE(true, true, P8_GELU);
        P8_CASE(true, true, P8_GATE);
        P8_CASE(true, true, P8_RESID);
        P8_CASE(true, true, P8_SLAB);
        default:
            if (L.ntw == 4) p8_launch_one<true, true, 4, P8_ANY>(L.a, L.lda, L.b, L.ldb, L.nk, L.tiles_m, L.tiles_n, L.split, L.grid, e, s, L.team_delay);
            else p8_launch_one<true, true, 3, P8_ANY>(L.a, L.lda, L.b, L.ldb, L.nk, L.tiles_m, L.tiles_n, L.split, L.grid, e, s, L.team_delay);
    }
}
