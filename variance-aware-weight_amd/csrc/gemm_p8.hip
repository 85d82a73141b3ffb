// Persistent 256-row-tile bf16 MFMA GEMM: shape planning and dispatch.  The kernel is in gemm_p8_kernel.h; its
// instantiations live in one translation unit per operand layout (gemm_p8_fwd / _dgrad / _wgrad .hip) so they build in parallel.
#include "gemm_p8_kernel.h"

void p8_launch_fwd(const P8Launch& L, const EpiDev& e, hipStream_t s);      // A [M][K], B [N][K]
void p8_launch_dgrad(const P8Launch& L, const EpiDev& e, hipStream_t s);    // A [M][K], B [K][N]
void p8_launch_wgrad(const P8Launch& L, const EpiDev& e, hipStream_t s);    // A [K][M], B [K][N]
void p8_launch_tn(const P8Launch& L, const EpiDev& e, hipStream_t s);       // A [K][M], B [N][K]

// ---- host side ----------------------------------------------------------------------------------------------
// Tile width and split count for a shape, or use = false when the 128 x 128 kernel of gemm.hip should keep it.
struct P8Plan {
    bool use;
    int ntw, split, grid;
};

// Workgroups a persistent launch may use = CUs - reserved.  A workgroup of this kernel owns a CU's whole register file and LDS
// and the items are partitioned statically, so a kernel of another stream that holds even 8 CUs (RCCL's all-reduce during a
// data-parallel backward) makes a full-width persistent grid run its last workgroups in a second pass: measured with a stand-in
// (tools/contention_bench.py) 14.3 -> 19.6 ms/step, against 16.0 when the persistent grids leave those CUs alone.
// vaw_p8_set_reserved_cus(n) (parallel.py: during the backward of a data-parallel step) / VAW_P8_RESERVE_CUS=n (always).
static int g_p8_reserved = 0;
extern "C" void vaw_p8_set_reserved_cus(int n) { g_p8_reserved = n > 0 ? n : 0; }
static int p8_num_cus() {
    static int n = 0, env_r = 0;
    if (!n) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
        const char* v = getenv("VAW_P8_RESERVE_CUS");
        env_r = v ? atoi(v) : 0;
    }
    const int r = env_r > g_p8_reserved ? env_r : g_p8_reserved;
    return (r > 0 && r < n - 8) ? n - r : n;
}

int vaw_p8_cus_available() { return p8_num_cus(); }

// Measured on MI355X (tools/gemm_bench.py --tile cmp, DiT-B/4 shapes, one process): per item the 192-column tile costs
// ~0.85x the 256-column one in multi-round launches (12 instead of 16 MFMAs per phase over the same barriers; 96-byte bf16
// row segments in its epilogue) and is no faster when every workgroup has one item only (N = 768: 192 vs 256 items), so
// 192 is chosen only where it saves whole rounds (N = 2304: 3 rounds of full-width work instead of 3 x 256-wide with
// the last one a quarter full ... measured 77 vs 82 us; N = 3072: 4 rounds vs 3, measured 132 vs 98 us).
static double p8_cost(int64_t M, int64_t N, int ntw, int split, int cus) {
    const int bn = 64 * ntw;
    const int64_t items = ((M + 255) / 256) * ((N + bn - 1) / bn) * split;
    const int64_t rounds = (items + cus - 1) / cus;
    return (double)rounds * (ntw == 4 ? 1.0 : 0.85) / split;
}

P8Plan vaw_p8_plan(int64_t M, int64_t N, int64_t K, bool plain_f32, bool want_colsum, int64_t ws_floats, int force) {
    P8Plan pl{false, 4, 1, 0};
    if (force == 0) return pl;
    const int cus = p8_num_cus();
    const int nk = (int)(K / 64);
    double best = 1e30;
    for (int ntw = 4; ntw >= 3; --ntw) {
        if (force == 2 && ntw != 4) continue;
        if (force == 3 && ntw != 3) continue;
        const int bn = 64 * ntw;
        const int64_t tiles = ((M + 255) / 256) * ((N + bn - 1) / bn);
        if (ntw == 3 && force != 3) {
            // 192-column tiles: where they cut the padded width (N = 192 k that is not a multiple of 256: the conv channel
            // counts 192 / 576 / 960 ...), save whole rounds of a multi-round launch, or (N = 768 at M = 16384: 256 tiles of
            // 0.85 instead of 192 tiles on 256 CUs -- measured 3-9 % faster on all five DiT-B/4 launches of that width) put
            // idle CUs to work within one round; the cost model below decides, wider padding is the only veto
            const int64_t pad4 = ((N + 255) / 256) * 256, pad3 = ((N + 191) / 192) * 192;
            if (pad3 > pad4) continue;
        }
        int smax = 1;
        if (plain_f32 && !want_colsum && ws_floats > 0) {
            int64_t s = nk / 4;                                  // every split keeps >= 4 K tiles (256 of K)
            if (s > ws_floats / (M * N)) s = ws_floats / (M * N);
            if (s > 64) s = 64;
            if (s > cus / tiles) s = cus / tiles;                // one round of split items at most
            smax = s < 1 ? 1 : (int)s;
        }
        for (int s = 1; s <= smax; ++s) {
            const int per = (nk + s - 1) / s;
            if ((nk + per - 1) / per != s) continue;             // no empty splits
            // K tiles of MFMA work on the slowest workgroup, plus ~3 K tiles' worth of slab traffic per split item
            double c = p8_cost(M, N, ntw, s, cus) * nk;
            if (s > 1) c += 3.0;
            if (c < best - 1e-9) { best = c; pl.ntw = ntw; pl.split = s; }
        }
    }
    const int bn = 64 * pl.ntw;
    const int64_t items = ((M + 255) / 256) * ((N + bn - 1) / bn) * pl.split;
    pl.grid = (int)(items < cus ? items : cus);
    // worth it when most of the chip gets a tile and the K loop is long enough to fill the two-stage ring
    pl.use = force > 0 || (items >= cus / 2 && nk / pl.split >= 4 && M >= 256 && N >= 128);
    return pl;
}

void vaw_p8_launch(const P8Plan& pl, int a_kmajor, int b_kmajor, int64_t M, int64_t N, int64_t K, const bf16_t* a, int64_t lda,
                   const bf16_t* b, int64_t ldb, const EpiDev& e, hipStream_t s) {
    const int bn = 64 * pl.ntw;
    const int tiles_m = (int)((M + 255) / 256), tiles_n = (int)((N + bn - 1) / bn), nk = (int)(K / 64);
    // team stagger (see the kernel): only when a workgroup runs at least two items; delay = VAW_P8_TEAM_PCT % of the
    // estimated main-loop time of one item (1.6 us per 256 x 256 x 64 K tile)
    static int team_pct = -1;
    if (team_pct < 0) { const char* v = getenv("VAW_P8_TEAM_PCT"); team_pct = v ? atoi(v) : 0; }
    const int64_t items = (int64_t)tiles_m * tiles_n * pl.split;
    const int nk_item = (nk + pl.split - 1) / pl.split;
    const int team_delay = (team_pct > 0 && items >= 2 * pl.grid) ? (int)(nk_item * 160LL * pl.ntw / 4 * team_pct / 100) : 0;
    P8Launch L{a, b, lda, ldb, nk, tiles_m, tiles_n, pl.split, pl.grid, pl.ntw, 0, team_delay};
    // epilogue kind (gemm_epi.h): the specialised kernels cover the launches of the training step, P8_ANY the rest
    const bool bf16_out = !e.out_f32;
    if (pl.split > 1) L.epi = P8_SLAB;
    else if (e.act == 1 && e.aux_out && !e.gate && !e.resid && !e.rowadd && bf16_out && !e.colpart) L.epi = P8_GELU;
    else if (e.act == 2 && !e.bias && !e.aux_out && !e.gate && !e.resid && !e.rowadd && bf16_out && e.alpha == 1.f) L.epi = P8_DGELU;
    else if (e.act == 0 && e.gate && e.resid && !e.resid_act && e.aux_out && !e.rowadd && e.out_f32 && e.beta == 0.f && !e.colpart)
        L.epi = P8_GATE;
    else if (e.act == 0 && !e.aux_out && !e.gate && !e.resid && !e.rowadd && e.beta == 0.f) L.epi = P8_STORE;
    else if (a_kmajor && b_kmajor && e.act == 0 && !e.aux_out && !e.gate && e.resid && e.resid_act && !e.rowadd && bf16_out && e.beta == 0.f && !e.colpart)
        L.epi = P8_RESID;           // 1x1 convs with a fused skip add (UNet attention proj_out): P8_ANY ran them at 2.6x the time of the plain store
    else L.epi = P8_ANY;
    if (a_kmajor && b_kmajor) p8_launch_fwd(L, e, s);
    else if (a_kmajor && !b_kmajor) p8_launch_dgrad(L, e, s);
    else if (!a_kmajor && !b_kmajor) p8_launch_wgrad(L, e, s);
    else p8_launch_tn(L, e, s);
}

// ---- grouped weight gradients -----------------------------------------------------------------------------------
#include <vector>

void vaw_p8_group_fp8(int ntw, bool a_e5m2, int nk, int grid, const EpiDev& e, const P8Prob* probs_dev, const P8Group& grp, hipStream_t s);   // gemm_p8_fp8.hip

extern "C" int64_t vaw_wgrad_grouped_desc_bytes(int n_problems) { return (int64_t)n_problems * (int64_t)sizeof(P8Prob); }

extern "C" int vaw_wgrad_grouped(vaw_dtype dt, int n_problems, const vaw_wgrad_problem* problems, int64_t K, float beta,
                                 void* desc_dev, int upload, float* workspace, int64_t workspace_floats, vaw_stream stream) {
    VAW_CHECK_ARG(dt == VAW_BF16 || dt == VAW_FP8 || dt == VAW_BF8, "wgrad_grouped: bf16 or fp8 operands (the f32 parity mode runs vaw_gemm per layer)");
    const bool f8 = dt == VAW_FP8 || dt == VAW_BF8;
    const int kt = f8 ? 128 : 64;
    VAW_CHECK_ARG(n_problems > 0 && problems && desc_dev && K > 0 && K % kt == 0, "wgrad_grouped: bad arguments (K %% %d)", kt);
    bool all192 = true, any_not256 = false;
    for (int i = 0; i < n_problems; ++i) {
        const vaw_wgrad_problem& q = problems[i];
        VAW_CHECK_ARG(q.dy && q.x && q.dw && q.M >= 16 && q.N >= 16 && q.M % 8 == 0 && q.N % 8 == 0, "wgrad_grouped: problem %d: M, N", i);
        if (f8) VAW_CHECK_ARG(q.ld_dy >= K && q.ld_x >= K && q.ld_dw >= q.N && q.ld_dy % 16 == 0 && q.ld_x % 16 == 0 && q.ld_dw % 4 == 0,
                              "wgrad_grouped: problem %d: leading dimensions (fp8: dy^T [M][K], x^T [N][K])", i);
        else VAW_CHECK_ARG(q.ld_dy >= q.M && q.ld_x >= q.N && q.ld_dw >= q.N && q.ld_dy % 8 == 0 && q.ld_x % 8 == 0 && q.ld_dw % 4 == 0,
                           "wgrad_grouped: problem %d: leading dimensions", i);
        VAW_CHECK_ARG((((uintptr_t)q.dy | (uintptr_t)q.x | (uintptr_t)q.dw) & 15) == 0, "wgrad_grouped: problem %d: alignment", i);
        VAW_CHECK_ARG(q.M < (1 << 30) && q.N < (1 << 30), "wgrad_grouped: problem %d too large", i);
        all192 = all192 && q.N % 192 == 0;
        any_not256 = any_not256 || q.N % 256 != 0;
    }
    const int ntw = (all192 && any_not256) ? 3 : 4, bn = 64 * ntw;
    static thread_local std::vector<P8Prob> host;
    host.resize(n_problems);
    int64_t t_total = 0;
    for (int i = 0; i < n_problems; ++i) {
        const vaw_wgrad_problem& q = problems[i];
        P8Prob& h = host[i];
        h.a = (const bf16_t*)q.dy; h.b = (const bf16_t*)q.x; h.c = q.dw;
        h.lda = q.ld_dy; h.ldb = q.ld_x; h.ldc = q.ld_dw;
        h.M = (int)q.M; h.N = (int)q.N;
        h.alpha = q.alpha != 0.f ? q.alpha : 1.f;
        h.pad_ = 0;
        h.scale_a = f8 ? q.scale_dy : nullptr;
        h.scale_b = f8 ? q.scale_x : nullptr;
        h.tiles_n = (int)((q.N + bn - 1) / bn);
        h.tile0 = (int)t_total;
        t_total += ((q.M + 255) / 256) * h.tiles_n;
        VAW_CHECK_ARG(t_total < (1 << 30), "wgrad_grouped: too many tiles");
    }
    hipStream_t s = (hipStream_t)stream;
    if (upload) {
        const hipError_t rc = vaw_upload_table(desc_dev, host.data(), sizeof(P8Prob) * n_problems, s);
        VAW_CHECK_ARG(rc == hipSuccess, "wgrad_grouped: descriptor upload failed: %s", hipGetErrorString(rc));
    }
    const int cus = p8_num_cus(), nk = (int)(K / kt);
    P8Group grp{};
    grp.n_prob = n_problems;
    grp.t_full = (int)(t_total / cus) * cus;
    grp.t_rem = (int)(t_total - grp.t_full);
    grp.n_split = 1;
    grp.slab = workspace;
    if (grp.t_rem > 0) {
        int64_t sp = cus / grp.t_rem;                               // the K-split tiles fill one more round of workgroups
        if (sp > nk / 4) sp = nk / 4;                               // every split keeps >= 4 K tiles
        const int64_t per_split = (int64_t)grp.t_rem * 256 * bn;
        if (workspace_floats <= 0 || !workspace) sp = 0;
        else if (sp > workspace_floats / per_split) sp = workspace_floats / per_split;
        if (sp < 2) {                                               // no room / too little K to split: whole tiles, a partial round
            grp.t_full = (int)t_total;
            grp.t_rem = 0;
        } else {
            const int per = (nk + (int)sp - 1) / (int)sp;
            grp.n_split = (nk + per - 1) / per;                     // no empty splits
        }
    }
    const int64_t items = grp.t_full + (int64_t)grp.t_rem * grp.n_split;
    const int grid = (int)(items < cus ? items : cus);
    EpiDev e{};
    e.alpha = 1.f;
    e.beta = beta;
    e.out_f32 = 1;
    e.rpb = 1;
    e.nt_off = 1;
    {
        static int dbg = -1;
        if (dbg < 0) { const char* v = getenv("VAW_GEMM_DEBUG"); dbg = v ? atoi(v) : 0; }
        e.debug = dbg;
    }
    if (f8) vaw_p8_group_fp8(ntw, dt == VAW_BF8, nk, grid, e, (const P8Prob*)desc_dev, grp, s);
    else if (ntw == 4) p8_launch_group<4>(nk, grid, e, (const P8Prob*)desc_dev, grp, s);
    else p8_launch_group<3>(nk, grid, e, (const P8Prob*)desc_dev, grp, s);
    if (grp.t_rem > 0) {
        if (ntw == 4) p8_group_fixup_kernel<256><<<grp.t_rem * 8, 256, 0, s>>>((const P8Prob*)desc_dev, grp, beta);
        else p8_group_fixup_kernel<192><<<grp.t_rem * 8, 256, 0, s>>>((const P8Prob*)desc_dev, grp, beta);
    }
    VAW_CHECK_LAUNCH("wgrad_grouped");
    return VAW_OK;
}
