/* vaw_hip.h -- C ABI of libvaw_hip.so: the MI355X (gfx950) kernels underneath the
 * variance-aware-weighted diffusion training step.
 *
 * The reference (LilYau350/Variance-Aware-Weight) has no FFI for this path: it is
 * Python calling ATen/cuDNN/cuBLAS.  The drop-in boundary is therefore its Python
 * call surface (Trainer.train_step -> GaussianDiffusion.training_losses -> model),
 * mirrored by the `vaw_amd` package, and THIS header is what that package binds
 * through ctypes.  Each entry point names the reference arithmetic it replaces
 * (paths are relative to the reference repository root).
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless its name ends in _host;
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *  - no allocation, no synchronisation, no hidden state: safe for stream capture;
 *  - return value: VAW_OK or a negative vaw_status; vaw_last_error_string() gives
 *    the message of the calling thread's last failure;
 *  - "act dtype" is the storage type of activations: VAW_F32 (parity mode) or
 *    VAW_BF16 (throughput mode).  Accumulation, statistics, the residual stream,
 *    gradients of parameters and the optimizer state are always f32.
 */
#ifndef VAW_HIP_H
#define VAW_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum { VAW_OK = 0, VAW_ERR_INVALID = -1, VAW_ERR_LAUNCH = -2, VAW_ERR_UNSUPPORTED = -3 } vaw_status;
typedef enum { VAW_F32 = 0, VAW_BF16 = 1,
               VAW_FP8 = 2, VAW_BF8 = 3 /* OCP e4m3fn / e5m2: GEMM operand formats of vaw_gemm_fp8 and vaw_wgrad_grouped only */ } vaw_dtype;
typedef void* vaw_stream;

int vaw_version(void);
const char* vaw_last_error_string(void);

/* ---------------------------------------------------------------------------
 * Diffusion objective  (tools/gaussian_diffusion.py)
 * ------------------------------------------------------------------------- */

/* q_sample :234-252 with the table gather of _extract_into_tensor :1059-1072 fused in:
 *   x_t[b,:] = tab_a[t[b]] * x0[b,:] + tab_s[t[b]] * noise[b,:]
 * tab_a/tab_s: f32[T] = float(float64 table).  t out of [0,T) poisons the row with NaN. */
int vaw_qsample_fwd(const float* x0, const float* noise, const int64_t* t, const float* tab_a, const float* tab_s,
                    int num_timesteps, float* x_t, int B, int64_t per_sample, vaw_stream stream);

/* out[b,:] = ca[b]*x[b,:] + cb[b]*y[b,:]  -- FlowMatching.q_sample :1277-1281 and the
 * VELOCITY / VECTOR targets of compute_target :818-832, :1284-1300. */
int vaw_mix_rows(const float* x, const float* y, const float* ca, const float* cb, float* out, int B,
                 int64_t per_sample, vaw_stream stream);

/* training_losses :908-913 fused: target = ca[b]*x0 + cb[b]*noise (compute_target),
 * mse[b] = w[b] * mean_flat((target - model_out)^2)  (tools/nn.py:86-90). */
int vaw_wmse_fwd(const float* model_out, const float* x0, const float* noise, const float* ca, const float* cb,
                 const float* w, float* mse, int B, int64_t per_sample, vaw_stream stream);
/* d(model_out)[b,:] = gmse[b] * w[b] * 2 * (model_out - target) / per_sample */
int vaw_wmse_bwd(const float* model_out, const float* x0, const float* noise, const float* ca, const float* cb,
                 const float* w, const float* gmse, float* dout, int B, int64_t per_sample, vaw_stream stream);

/* Variational-bound term of the learned-variance objective and the KL losses: _vb_terms_bpd
 * (gaussian_diffusion.py:775-808) = q_posterior_mean_variance :254-276 + the training side of p_mean_variance
 * :278-384 + normal_kl / discretized_gaussian_log_likelihood (tools/losses.py:12-76) + mean_flat / ln 2, fused.
 *   vb[b] = scale * mean_flat(t[b]==0 ? decoder NLL : KL(q(x_{t-1}|x_t,x_0) || p(x_{t-1}|x_t))) / ln 2
 * coef: f32 [B][8] per-sample table rows {posterior_mean_coef1, posterior_mean_coef2, posterior_log_variance_clipped,
 * log(beta_t) (LEARNED_RANGE upper end) or the fixed model log variance, pa, pb (pred_xstart = pa*x_t + pb*mean_out),
 * t==0 ? 1 : 0, unused}.  mean_mode 0: model mean = posterior mean of pred_xstart, 1: mean_out itself (PREVIOUS_X).
 * var_mode 0: fixed, 1: LEARNED (var_out = log variance), 2: LEARNED_RANGE (var_out in [-1,1]).  All tensors f32,
 * [B, per_sample] contiguous.  bwd: d_var / d_mean (either may be NULL: the MSE+vb objective detaches the mean). */
int vaw_vb_fwd(const float* mean_out, const float* var_out, const float* x0, const float* x_t, const float* coef,
               int mean_mode, int var_mode, float scale, float* vb, int B, int64_t per_sample, vaw_stream stream);
int vaw_vb_bwd(const float* mean_out, const float* var_out, const float* x0, const float* x_t, const float* coef,
               int mean_mode, int var_mode, float scale, const float* gvb, float* d_mean, float* d_var, int B,
               int64_t per_sample, vaw_stream stream);

/* One reverse-process step of the sampling side, fused: p_mean_variance (gaussian_diffusion.py:278-384, no
 * denoised_fn / cond_fn) followed by p_sample (:461-505, kind 1) or ddim_sample (:603-651, kind 2); kind 0 only
 * fills pred_xstart / mean / log_variance.  coef: f32 [B][16] per-sample rows of the timestep tables (layout in
 * csrc/elementwise.hip; built by vaw_amd.GaussianDiffusion._sample_table with the reference's f64 -> f32 casts).
 * noise: the randn_like(x) draw of the step (caller's RNG).  Outputs may be NULL.  All f32 [B, per_sample]. */
int vaw_sample_step(int kind, const float* mean_out, const float* var_out, const float* x, const float* noise,
                    const float* coef, int mean_mode, int var_mode, int clip_denoised, float eta, float* sample,
                    float* pred_xstart, float* mean, float* log_variance, int B, int64_t per_sample, vaw_stream stream);

/* ---------------------------------------------------------------------------
 * Dense layers  (nn.Linear / Conv2d(k=p,s=p) / Conv1d(k=1) in models/dit.py, models/unet.py;
 * cuBLAS in the reference).  One GEMM entry point, MFMA inside.
 * ------------------------------------------------------------------------- */

typedef struct {
    const float* bias;     /* f32[N] added to every row, or NULL */
    int act;               /* 0 none | 1 GELU(tanh) forward | 2 multiply by GELU'(aux_in) (backward) */
    const void* aux_in;    /* act dtype [M,N] (ld = ldc): pre-activation for act==2 */
    void* aux_out;         /* act dtype [M,N] (ld = ldc): value BEFORE act/gate/residual is stored here, or NULL */
    const float* gate;     /* f32: gate[(m / rows_per_batch) * gate_ld + n], or NULL */
    int64_t gate_ld;
    const void* resid;     /* [M,N] (ld = ldc) residual added AFTER the gate, or NULL; f32 unless resid_is_act */
    const float* rowadd;   /* f32 [rows_per_batch, N] added by (m % rows_per_batch): frozen pos_embed, or NULL */
    int rows_per_batch;    /* tokens per sample (T); required when gate/rowadd is set */
    float alpha;           /* scales the accumulator first */
    float beta;            /* C = beta*C_old + value (f32 output only; 0 = overwrite, C_old not read) */
    int out_f32;           /* 1: C is f32 regardless of act dtype; 0: C has act dtype */
    float* colsum_out;     /* f32[N] or NULL: colsum_out = colsum_beta*colsum_out + sum_m C[m,:] (values as stored),
                            * i.e. the bias gradient of the layer whose output gradient this GEMM produces; taken in
                            * the epilogue (no second pass over C); needs the workspace */
    float colsum_beta;
    int resid_is_act;      /* 1: resid has the act dtype (UNet skip connections), 0: f32 (DiT residual stream) */
    float* rowsum_a_out;   /* f32[M] or NULL: rowsum_a_out = rowsum_a_beta*rowsum_a_out + sum_k op(A)[m,k].  For the weight
                            * gradient dW = dy^T x (a_kmajor = 0) this is the layer's BIAS gradient sum_rows dy, taken from
                            * the dy tiles the MFMA kernel stages anyway (one extra MFMA against a ones operand, fixed
                            * order); other layouts / the generic kernel honour it with a separate pass.  Needs the workspace */
    float rowsum_a_beta;
    float* colsum_partial_out;  /* deferred form of colsum_out (which must then be NULL): the kernel leaves its per-row-tile partial
                                 * column sums of C, [*colsum_rows_out][N] f32 (at most ceil(M/64) rows: size the buffer for that),
                                 * here and does NOT fold them; the caller folds many such buffers in one launch
                                 * (vaw_reduce_rows_batched).  No workspace needed for it */
    int64_t* colsum_rows_out;   /* HOST address, in / out (required with the above): on entry the capacity of colsum_partial_out in
                                 * rows (a launch that needs more fails with VAW_ERR_INVALID before anything runs), on return
                                 * the number of partial rows written */
} vaw_epilogue;

/* C[M,N] = epilogue( alpha * op(A)[M,K] . op(B)[K,N] )
 *   a_kmajor=1: A stored [M][K] (lda = row stride);  0: stored [K][M]
 *   b_kmajor=1: B stored [N][K] (ldb = row stride);  0: stored [K][N]
 * forward  y = x W^T + b : (1,1)   dgrad dx = dy W : (1,0)   wgrad dW = dy^T x : (0,0)
 * A and B are act dtype.  value = acc*alpha + bias -> [aux_out] -> act -> *gate -> +resid -> +rowadd.
 * workspace (f32, may be NULL): lets launches with a plain f32 epilogue (weight gradients: K = B*T, few output
 * tiles) run split-K -- partial slabs [split][M][N] summed in a fixed order by a second kernel, so results do
 * not depend on scheduling.  A workspace of 8*M*N floats is always enough; smaller ones lower the split. */
int vaw_gemm(vaw_dtype dt, int a_kmajor, int b_kmajor, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda,
             const void* B, int64_t ldb, void* C, int64_t ldc, const vaw_epilogue* epi_host, float* workspace,
             int64_t workspace_floats, vaw_stream stream);

/* Weight gradients of many Linear layers in ONE launch: dW_p[M_p,N_p] = beta*dW_p + dy_p[K,M_p]^T . x_p[K,N_p] for
 * p = 0..n_problems-1, all sharing K (= tokens of the batch).  Replaces the per-layer `dy^T x` GEMMs that autograd
 * issues in the reference (models/dit.py:118-155 backward of qkv / proj / fc1 / fc2), deferred to the end of backward:
 * with K = B*T long and M_p*N_p small, a per-layer launch must split K over the chip and move every tile through an
 * f32 slab; all layers together give enough whole tiles for every CU, and only the tiles of the last, partial round are
 * K-split (folded in a fixed order: results do not depend on scheduling).
 * problems: HOST array; desc_dev: device scratch of vaw_wgrad_grouped_desc_bytes(n) bytes that holds the device copy
 * of the table -- upload != 0 (re)writes it on `stream` (pass 1 the first time and whenever a pointer changed).
 * workspace: f32 slabs for the K-split tiles (256 * 256 * n_CUs floats are always enough; smaller ones lower the split). */
typedef struct {
    const void* dy;        /* act dtype (bf16) [K][M], row stride ld_dy */
    const void* x;         /* act dtype (bf16) [K][N], row stride ld_x */
    float* dw;             /* f32 [M][N], row stride ld_dw */
    int64_t M, N, ld_dy, ld_x, ld_dw;
    float alpha;           /* scales dy^T x before it is added (0 = 1) */
    int pad_;
    const float* scale_dy; /* dt = VAW_FP8 / VAW_BF8 only: dy and x are the TRANSPOSED e4m3 copies dy^T [M][K] and x^T [N][K] (k-major), ld_*  */
    const float* scale_x;  /* their row strides, and these DEVICE scalars their per-tensor dequantisation scales (vaw_fp8_quantize) */
} vaw_wgrad_problem;
int64_t vaw_wgrad_grouped_desc_bytes(int n_problems);
int vaw_wgrad_grouped(vaw_dtype dt, int n_problems, const vaw_wgrad_problem* problems, int64_t K, float beta, void* desc_dev,
                      int upload, float* workspace, int64_t workspace_floats, vaw_stream stream);

/* fp8 operands (OCP e4m3fn, per-tensor scaling) for the Linear GEMMs: BASELINE.json config 5 "DiT-XL/2, fp8 MFMA GEMMs + bf16
 * accum" (model: models/dit.py:373, recipe run.sh:20-26; the reference itself trains that model in bf16 autocast).
 *
 * vaw_fp8_quantize: q[r,c] = fp8(src[r,c] * FMAX / amax|src|) (round to nearest even), qt[c,r] = q[r,c] (optional transposed
 * copy: every fp8 GEMM takes both operands k-major, so dgrad reads W^T and wgrad reads dy^T and x^T), and *scale_out =
 * amax / FMAX (1 for an all-zero tensor), a DEVICE scalar: nothing syncs with the host.  dst_format: VAW_FP8 (e4m3fn, FMAX 448:
 * weights and activations) or VAW_BF8 (e5m2, FMAX 57344: gradients).  src: f32 or bf16 [R][C], row stride ld; q: [R][C] bytes,
 * stride ldq; qt: [C][R] bytes, stride ldt.  C, ld, ldq, ldt multiples of 4.
 * workspace: vaw_fp8_quantize_workspace_floats() floats (amax partials, folded in a fixed order). */
int64_t vaw_fp8_quantize_workspace_floats(void);
int vaw_fp8_quantize(vaw_dtype src_dt, vaw_dtype dst_format, const void* src, int64_t R, int64_t C, int64_t ld, void* q, int64_t ldq,
                     void* qt, int64_t ldt, float* scale_out, float* workspace, int64_t workspace_floats, vaw_stream stream);
/* Delayed scaling (the step after the first): one pass instead of two.  state = 4 floats {scale in use, running max |src|,
 * FMAX / margin, unused}: vaw_fp8_quantize_delayed quantises with state[0] as it stands (values beyond its range saturate) and
 * folds this tensor's max |src| into state[1] (integer atomic max on the bits: order-independent); vaw_fp8_scale_update, once
 * per step over all n states ([n][4] floats), turns every non-zero state[1] into the next scale state[0] = state[1] / state[2]
 * and clears it.  GEMMs take &state[0] as their scale pointer. */
int vaw_fp8_quantize_delayed(vaw_dtype src_dt, vaw_dtype dst_format, const void* src, int64_t R, int64_t C, int64_t ld, void* q,
                             int64_t ldq, void* qt, int64_t ldt, float* state, vaw_stream stream);
int vaw_fp8_scale_update(float* states, int64_t n, vaw_stream stream);
/* vaw_fp8_quantize_delayed for MANY f32 tensors in one launch (e4m3): the once-per-step re-quantisation of all Linear weights from
 * their f32 masters (the reference's autocast casts the same weights to bf16 on every use, tools/trainer.py:104-108).  jobs: host
 * array; desc_dev: device buffer of vaw_fp8_quantize_batched_desc_bytes(n_jobs) bytes, written when upload != 0 (addresses are
 * static between steps).  Bytes, transposed copies and running maxima equal the per-tensor calls'. */
typedef struct {
    const float* src;      /* f32 [R][C], row stride ld */
    void* q;               /* e4m3 bytes [R][C], row stride ldq */
    void* qt;              /* e4m3 bytes [C][R], row stride ldt, or NULL */
    float* state;          /* the tensor's delayed-scaling state (4 floats) */
    int64_t R, C, ld, ldq, ldt;
} vaw_fp8_quant_job;
int64_t vaw_fp8_quantize_batched_desc_bytes(int n_jobs);
int vaw_fp8_quantize_delayed_batched(int n_jobs, const vaw_fp8_quant_job* jobs, void* desc_dev, int upload, vaw_stream stream);
/* C[M,N] = epilogue(alpha * *scale_a * *scale_b * A[M,K] . B[N,K]^T): A bytes of a_format (VAW_FP8 | VAW_BF8), B e4m3 bytes,
 * both k-major (K % 128 == 0, row strides multiples of 16), f32 accumulation on v_mfma_scale_f32_16x16x128_f8f6f4 with unit
 * block scales; C and the epilogue operands as for vaw_gemm with dt = VAW_BF16 (bf16 C / aux, or f32 C with out_f32).
 * Epilogues offered: bias (+ colsum_out), either A format; GELU' (+ colsum_out), either; bias + aux_out + GELU and bias +
 * aux_out + gate + f32 residual, e4m3 A.  Others return VAW_ERR_UNSUPPORTED.
 * Weight gradients: vaw_wgrad_grouped with dt = VAW_FP8 (dy^T, x^T e4m3) or VAW_BF8 (dy^T e5m2, x^T e4m3).
 * c_fp8_state != NULL (GELU and GELU' epilogues only): C is written as fp8 BYTES of c_fp8_format (row stride ldc bytes) -- the bf16
 * rounding of each value, divided by the scale in c_fp8_state[0] and saturated, exactly what vaw_fp8_quantize_delayed would make
 * of the bf16 tensor -- and the tensor's max |x| is folded into c_fp8_state[1]; vaw_fp8_transpose then provides the transposed
 * copy.  For fp8 mode's fc1 output and fc2 input gradient, whose bf16 forms have no other reader. */
int vaw_gemm_fp8(vaw_dtype a_format, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const float* scale_a,
                 const void* B, int64_t ldb, const float* scale_b, void* C, int64_t ldc, const vaw_epilogue* epi_host,
                 float* c_fp8_state, vaw_dtype c_fp8_format, float* workspace, int64_t workspace_floats, vaw_stream stream);
/* qt[c][r] = q[r][c] for fp8 bytes; R % 64 == 0, C % 128 == 0, ldq % 8 == 0, ldt % 4 == 0. */
int vaw_fp8_transpose(const void* q, int64_t R, int64_t C, int64_t ldq, void* qt, int64_t ldt, vaw_stream stream);

/* Keep n CUs out of every persistent-GEMM grid from now on (0 = none): for the time a kernel of another stream (a collective)
 * holds CUs of its own -- a persistent workgroup cannot share its CU, and a grid that does not fit runs a second pass. */
void vaw_p8_set_reserved_cus(int n);
/* 1 when vaw_gemm would run these operands on the bf16 MFMA kernel (M%128==0, N%128==0, K%64==0, 16-byte
 * aligned rows), 0 when it takes the exact-f32 generic kernel.  For measurement and tests. */
int vaw_gemm_uses_bf16_mfma(vaw_dtype dt, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B,
                            int64_t ldb);

/* out[n] = beta*out[n] + sum_m X[m,n]  (bias gradients). X act dtype, out f32.  Two fixed-order stages through a
 * caller-provided f32 workspace of vaw_colsum_workspace_floats(M,N) elements: no atomics, bitwise reproducible. */
int64_t vaw_colsum_workspace_floats(int64_t M, int64_t N);
int vaw_colsum(vaw_dtype dt, const void* X, int64_t M, int64_t N, int64_t ldx, float* out, float beta,
               float* workspace, int64_t workspace_floats, vaw_stream stream);
/* out[n] = beta*out[n] + sum_{r<R} partial[r,n], r ascending (second stage of the fixed-order column sums) */
int vaw_reduce_rows(const float* partial, int64_t R, int64_t N, float* out, float beta, vaw_stream stream);
/* The same fold for MANY (partial, out) pairs in one launch: out_j[n] = beta*out_j[n] + sum_{r<R_j} partial_j[r,n], each with
 * vaw_reduce_rows' summation tree (bitwise the same results).  Used for the bias gradients of all Linear layers of a group of
 * DiT blocks (autograd of `x W^T + b`, models/dit.py:126-137: db = sum_rows dy), whose partial rows the dy-producing kernels
 * leave behind: one launch per group instead of one per layer.  jobs: host array; desc_dev: device buffer of
 * vaw_reduce_rows_batched_desc_bytes(n_jobs) bytes that receives the table when upload != 0 (addresses are static between
 * steps: upload once). */
typedef struct {
    const float* partial;  /* f32 [R][N] */
    float* out;            /* f32 [N] */
    int64_t R, N;
} vaw_reduce_job;
int64_t vaw_reduce_rows_batched_desc_bytes(int n_jobs);
int vaw_reduce_rows_batched(int n_jobs, const vaw_reduce_job* jobs, float beta, void* desc_dev, int upload, vaw_stream stream);

/* ---------------------------------------------------------------------------
 * DiT pieces  (models/dit.py)
 * ------------------------------------------------------------------------- */

/* modulate(LayerNorm(x), shift, scale) :24-25,125,135: eps=1e-6, no affine.  x: f32 [B*T, D] residual
 * stream; shift/scale: f32 rows of the adaLN output with row stride mod_ld; out: act dtype [B*T, D];
 * mean/rstd: f32 [B*T] saved for backward. */
int vaw_ln_modulate_fwd(vaw_dtype dt, const float* x, const float* shift, const float* scale, int64_t mod_ld,
                        void* out, float* mean, float* rstd, int B, int T, int D, float eps, vaw_stream stream);
/* Backward of the above, fused with the residual-stream gradient:
 *   dx[b,t,:]   = (dres_in ? dres_in : 0) + LN'(dout * (1+scale))
 *   dshift[b,:] = sum_t dout ;  dscale[b,:] = sum_t dout * xhat     (rows with stride dmod_ld)
 * dx may alias dres_in. */
int vaw_ln_modulate_bwd(vaw_dtype dt, const void* dout, const float* x, const float* mean, const float* rstd,
                        const float* scale, int64_t mod_ld, const float* dres_in, float* dx, float* dshift,
                        float* dscale, int64_t dmod_ld, int B, int T, int D, float* workspace, int64_t workspace_floats,
                        vaw_stream stream);
/* fp8 mode of the DiT blocks (delayed scaling): the same two kernels with their activation output written as fp8 BYTES [B*T][D]
 * -- the bf16 rounding of each value divided by q_state[0], saturated: what vaw_fp8_quantize_delayed makes of the bf16 tensor,
 * which then is never written -- and the tensor's max |x| folded into q_state[1]; vaw_fp8_transpose supplies the transposed copy. */
int vaw_ln_modulate_fwd_fp8(const float* x, const float* shift, const float* scale, int64_t mod_ld, void* q_out, float* q_state,
                            vaw_dtype q_format, float* mean, float* rstd, int B, int T, int D, float eps, vaw_stream stream);
int vaw_gate_bwd_fp8(const float* dres, const void* y, const float* gate, int64_t mod_ld, void* dy_q, float* q_state,
                     vaw_dtype q_format, float* dgate, int64_t dmod_ld, float* dy_colsum_partial, int B, int T, int D,
                     float* workspace, int64_t workspace_floats, vaw_stream stream);
/* Workspace (f32) of vaw_ln_modulate_bwd / vaw_gate_bwd: with it, a sample's T rows are cut into chunks handled by
 * separate workgroups (small per-GPU batches would otherwise leave most CUs idle: one workgroup per sample) and the
 * per-sample column sums are folded over the chunks in a fixed order by a second kernel.  NULL = one workgroup per sample. */
int64_t vaw_row_bwd_workspace_floats(int B, int T, int D);
/* Backward of `x + gate.unsqueeze(1) * y` :135-136 w.r.t. the branch:
 *   dy[b,t,:] = dres[b,t,:] * gate[b,:] (act dtype) ; dgate[b,:] = sum_t dres * y ;
 *   dy_colsum_partial (f32 [B, D] or NULL): per-sample sum_t dy -- reduce over B with vaw_reduce_rows to get the
 *   bias gradient of the branch's last Linear without re-reading dy. */
int vaw_gate_bwd(vaw_dtype dt, const float* dres, const void* y, const float* gate, int64_t mod_ld, void* dy,
                 float* dgate, int64_t dmod_ld, float* dy_colsum_partial, int B, int T, int D, float* workspace, int64_t workspace_floats, vaw_stream stream);
/* vaw_ln_modulate_bwd and the vaw_gate_bwd that consumes its dx, as ONE pass over the rows (autograd of models/dit.py:135-136
 * walked backwards: the LayerNorm+modulate of a branch, then the gated residual add in front of it): dx as vaw_ln_modulate_bwd;
 * dy_next = dx * gate_next (act dtype), dgate_next[b,:] = sum_t dx * y_next, dy_colsum_partial [B,D] = per-sample sum_t dy_next.
 * dgate_next shares dmod_ld with dshift / dscale.  Saves the re-read of dx (f32) and a launch; bitwise equal to the pair. */
int vaw_ln_modulate_bwd_gate(vaw_dtype dt, const void* dout, const float* x, const float* mean, const float* rstd,
                             const float* scale, int64_t mod_ld, const float* dres_in, float* dx, float* dshift, float* dscale,
                             int64_t dmod_ld, const void* y_next, const float* gate_next, void* dy_next, float* dgate_next,
                             float* dy_colsum_partial, int B, int T, int D, float* workspace, int64_t workspace_floats, vaw_stream stream);
/* fp8 mode: dy_next as fp8 bytes of its bf16 roundings (see vaw_gate_bwd_fp8) */
int vaw_ln_modulate_bwd_gate_fp8(const void* dout, const float* x, const float* mean, const float* rstd, const float* scale,
                                 int64_t mod_ld, const float* dres_in, float* dx, float* dshift, float* dscale, int64_t dmod_ld,
                                 const void* y_next, const float* gate_next, void* dy_q, float* q_state, vaw_dtype q_format,
                                 float* dgate_next, float* dy_colsum_partial, int B, int T, int D, float* workspace,
                                 int64_t workspace_floats, vaw_stream stream);

/* timm PatchEmbed (dit.py:192) input side: x f32 [B,C,H,W] -> tokens act dtype [B*(H/p)*(W/p), C*p*p],
 * column order (c, i, j) = Conv2d weight flattening. */
int vaw_patchify(vaw_dtype dt, const float* img, void* tok, int B, int C, int H, int W, int p, vaw_stream stream);
/* gradient of patchify w.r.t. the image: dtok f32 [B*h*w, C*p*p] -> dimg f32 [B,C,H,W] */
int vaw_patchify_bwd(const float* dtok, float* dimg, int B, int C, int H, int W, int p, vaw_stream stream);
/* unpatchify :243-256: tokens f32 [B*h*w, p*p*C] (column order (i, j, c)) -> img f32 [B,C,h*p,w*p];
 * vaw_unpatchify_bwd is the transpose map (dimg -> token rows, act dtype). */
int vaw_unpatchify(vaw_dtype dt, const float* tok, float* img, int B, int C, int H, int W, int p,
                   vaw_stream stream);
int vaw_unpatchify_bwd(vaw_dtype dt, const float* dimg, void* dtok, int B, int C, int H, int W, int p,
                       vaw_stream stream);

/* sinusoidal embedding (dit.py:56-74 == tools/nn.py:103-121): out[b] = [cos(t f_i) | sin(t f_i)], act dtype. */
int vaw_timestep_embedding(vaw_dtype dt, const float* t, void* out, int B, int dim, float max_period,
                           vaw_stream stream);
/* SiLU on f32 input; out act dtype.  bwd: dx(f32) += / = dy * silu'(x). */
int vaw_silu_fwd(vaw_dtype dt, const float* x, void* out, int64_t n, vaw_stream stream);
int vaw_silu_bwd(const float* x, const float* dy, float* dx, int64_t n, vaw_stream stream);
/* out[b,:] = a[b,:] + table[idx[b],:]  (c = t_emb + y_emb, dit.py:267-269 / unet.py:676) */
int vaw_add_embedding(const float* a, const float* table, const int64_t* idx, float* out, int B, int D,
                      int num_rows, vaw_stream stream);
/* nn.Embedding backward: dtable[r,:] = beta*dtable[r,:] + sum_{b: idx[b]==r} dc[b,:] (b ascending; every row
 * of the table is written, no atomics, no pre-zeroing) */
int vaw_embedding_bwd(const float* dc, const int64_t* idx, float* dtable, int B, int D, int num_rows, float beta,
                      vaw_stream stream);

/* Multi-head attention, softmax(scale * q k^T) v, strided so both layouts of the reference are served:
 *   DiT (timm Attention, dit.py:126): qkv [B*T, 3*H*hd] token-major -> stride_t = 3*H*hd, stride_d = 1
 *   UNet QKVAttention (unet.py:362-390): qkv [B, 3*H*ch, T] channel-major -> stride_t = 1, stride_d = T
 * q/k/v/o point at element (b=0, h=0, t=0, d=0) of each operand; stride_b / stride_h in elements.
 * lse: f32 [B*H*T] (row log-sum-exp, saved for backward). */
typedef struct {
    int B, H, T, hd;
    int64_t q_sb, q_sh, q_st, q_sd; /* shared by q, k, v (same tensor, different base) */
    int64_t o_sb, o_sh, o_st, o_sd;
    float scale;
} vaw_attn_desc;
int vaw_attn_fwd(vaw_dtype dt, const vaw_attn_desc* d_host, const void* q, const void* k, const void* v, void* o,
                 float* lse, vaw_stream stream);
/* dq/dk/dv use the q strides, d_o the o strides.  delta: f32 [B*H*T] workspace. */
int vaw_attn_bwd(vaw_dtype dt, const vaw_attn_desc* d_host, const void* q, const void* k, const void* v,
                 const void* o, const void* d_o, const float* lse, float* delta, void* dq, void* dk, void* dv,
                 vaw_stream stream);
/* vaw_attn_bwd that also leaves the column sums of dq | dk | dv behind as PARTIAL rows -- the bias gradient of the qkv Linear in
 * front of the attention (timm Attention, models/dit.py:126: db = sum over tokens of dqkv) without a second pass over dqkv:
 * colsum_partial [*rows_out][3*H*hd] f32 in packed-qkv column order [3][H][hd]; at most B*T/64 rows: *rows_out is in / out --
 * on entry the capacity of the buffer in rows (checked), on return the rows written; fold them with vaw_reduce_rows / vaw_reduce_rows_batched.  Only the bf16 MFMA kernels offer it (token-major or any
 * layout they accept): VAW_ERR_UNSUPPORTED, nothing launched, otherwise -- use vaw_attn_bwd + vaw_colsum then. */
int vaw_attn_bwd_colsum(vaw_dtype dt, const vaw_attn_desc* d, const void* q, const void* k, const void* v, const void* o,
                        const void* d_o, const float* lse, float* delta, void* dq, void* dk, void* dv, float* colsum_partial,
                        int64_t* rows_out, vaw_stream stream);

/* ---------------------------------------------------------------------------
 * UNet pieces  (models/unet.py, tools/nn.py) -- activations are NHWC: [B*H*W pixels, C channels], act dtype
 * ------------------------------------------------------------------------- */

/* conv3x3 (stride 1, pad 1, NHWC) with a narrow side of at most 4 channels: the 3-channel stem and output convs of
 * models/unet.py:492 (conv_nd(dims, in_channels, ch, 3, padding=1)) and :625 (zero_module(conv_nd(dims, input_ch, out_channels, 3, padding=1))).
 * Direct f32-accumulating kernels (K = 27..36 is too short for the MFMA tile).  w has the activation dtype.
 *   mode 0  forward, narrow input:    in [M][Cn]      -> out [M][Cw],  w = [Cw][3][3][Cn], bias [Cw] or NULL
 *   mode 1  input gradient of a conv with a narrow OUTPUT: in = dy [M][Cn] -> out = dx [M][Cw], w = [Cn][3][3][Cw]
 *   mode 2  forward, narrow output:   in [M][Cw]      -> out [M][Cn],  w = [Cn][3][3][Cw], bias [Cn] or NULL
 * Returns VAW_ERR_UNSUPPORTED (nothing launched) when Cn > 4 or Cw % 8 != 0. */
int vaw_conv3x3_narrow(vaw_dtype dt, int mode, const void* in, const void* w, const float* bias, void* out, int B, int H,
                       int W, int Cn, int Cw, vaw_stream stream);

/* GroupNorm32 (tools/nn.py:17-19,93-100; G groups, eps) fused with what follows it in ResBlock._forward
 * (unet.py:236-256):  y = act( GN(x)*gamma + beta [ *(1 + scale[b,c]) + shift[b,c] ] ), act = SiLU if silu else id.
 * scale/shift: f32 rows with stride film_ld (the emb_layers output), or both NULL.  mean/rstd: f32 [B*G] saved.
 * workspace: vaw_groupnorm_workspace_floats(B, HW, C) f32. */
int64_t vaw_groupnorm_workspace_floats(int B, int HW, int C);
int vaw_groupnorm_fwd(vaw_dtype dt, const void* x, const float* gamma, const float* beta, const float* scale,
                      const float* shift, int64_t film_ld, int silu, void* y, float* mean, float* rstd, int B, int HW,
                      int C, int G, float eps, float* workspace, vaw_stream stream);
/* dx = GN'(...) (+ dx_add, act dtype, may be NULL); dgamma/dbeta = grad_beta*old + sum; dscale/dshift rows (stride
 * dfilm_ld) when FiLM.  Fixed-order reductions. */
int vaw_groupnorm_bwd(vaw_dtype dt, const void* dout, const void* x, const float* mean, const float* rstd,
                      const float* gamma, const float* beta, const float* scale, const float* shift, int64_t film_ld,
                      int silu, const void* dx_add, void* dx, float* dgamma, float* dbeta, float grad_beta, float* dscale,
                      float* dshift, int64_t dfilm_ld, int B, int HW, int C, int G, float* workspace, vaw_stream stream);
/* conv3x3 stride 1 pad 1 as GEMM (round 1: explicit patch matrix).  col[m, tap*C + c] = x[pixel(m)+tap offset, c];
 * vaw_col2im3x3 is the transposed map written as a gather (deterministic): the input gradient from d(col). */
int vaw_im2col3x3(vaw_dtype dt, const void* x, void* col, int B, int H, int W, int C, vaw_stream stream);
int vaw_col2im3x3(vaw_dtype dt, const void* dcol, void* dx, int B, int H, int W, int C, vaw_stream stream);
/* conv3x3 (stride 1, pad 1) as IMPLICIT GEMM on the bf16 MFMA kernel: the patch matrix is never written; padding taps
 * read a zero page.  mode 0: out[M,Co] = conv(act=x[M,Ci]; w) with the vaw_gemm epilogue (bias, residual, column sums);
 * mode 1: out = dx[M,Ci] from act = dy[M,Co]; mode 2: out = dW[Co][9][Ci] f32 = beta*dW + dy^T . patches(x) with
 * act = dy, act2 = x (split-K through the workspace); in mode 2 ep->rowsum_a_out, if set, receives the conv's BIAS
 * gradient rowsum_a_beta*old + sum over pixels of dy[.,co] (taken from the dy tiles already in LDS, fixed order).
 * w: [Co][3][3][Ci] act dtype.  Returns VAW_ERR_UNSUPPORTED
 * (nothing launched) for shapes that need the explicit vaw_im2col3x3 + vaw_gemm path: f32, Ci or Co not a multiple
 * of 64 (mode 0 / 1), ... */
int vaw_conv3x3(vaw_dtype dt, int mode, const void* act, const void* act2, const void* w, void* out, int B, int H, int W,
                int Ci, int Co, const vaw_epilogue* epi_host, float* workspace, int64_t workspace_floats, vaw_stream stream);
/* conv3x3 weight gradient for the 3-channel stem / output convs (Ci <= 4 or Co <= 4), where a GEMM would be 3 wide:
 * dw[co][tap][ci] (f32, channels-last weight layout) = beta*dw + sum_m dy[m,co] * x[pixel(m)+tap, ci]; chunk
 * partials in the workspace are folded in a fixed order. */
int64_t vaw_conv3x3_wgrad_small_workspace_floats(int B, int H, int W, int Ci, int Co);
int vaw_conv3x3_wgrad_small(vaw_dtype dt, const void* dy, const void* x, float* dw, float beta, int B, int H, int W, int Ci,
                            int Co, float* workspace, int64_t workspace_floats, vaw_stream stream);
/* mode 0: out[Ho,Wo] = s * sum of the 2x2 block of in[2Ho,2Wo] (avg_pool2d with s=1/4; nearest-upsample^T with s=1)
 * mode 1: out[Ho,Wo] = s * in[Ho/2,Wo/2]                        (nearest x2 with s=1; avg_pool2d^T with s=1/4) */
int vaw_resample2(vaw_dtype dt, const void* in, void* out, int B, int Ho, int Wo, int C, int mode, float s,
                  vaw_stream stream);
/* cat[m,:] = [a[m,:Ca] | b[m,:Cb]] (torch.cat(dim=1) of unet.py:684 in NHWC); split=1 copies cat back into a and b */
int vaw_concat_channels(vaw_dtype dt, void* a, void* b, void* cat, int64_t M, int Ca, int Cb, int split,
                        vaw_stream stream);
int vaw_add_inplace(vaw_dtype dt, void* dst, const void* src, int64_t n, vaw_stream stream);
/* ResBlock / resampling variants that no reference factory uses but its constructor accepts (models/unet.py:81-140,206-256):
 *   vaw_mul         out = a * b (act dtype): nn.Dropout(p) as a product with a pre-scaled keep mask (--dropout, main.py:99)
 *   vaw_subsample2  mode 0: out[b,i,j,:] = in[b,2i,2j,:] -- Downsample's stride-2 conv = the stride-1 conv sampled at the even
 *                   pixels; mode 1: its transpose (zeros at the odd pixels).  out is [B,Ho,Wo,C] in both modes
 *   vaw_rowvec_add  h[b,p,:] += e[b,:] (f32 e, row stride ld): `h + emb_out` of use_scale_shift_norm=False
 *   vaw_rowvec_sum  de[b,:] = beta*de[b,:] + sum_p dh[b,p,:]: its backward (fixed-order reduction) */
int vaw_mul(vaw_dtype dt, const void* a, const void* b, void* out, int64_t n, vaw_stream stream);
int vaw_subsample2(vaw_dtype dt, const void* in, void* out, int B, int Ho, int Wo, int C, int mode, vaw_stream stream);
int vaw_rowvec_add(vaw_dtype dt, void* h, const float* e, int64_t ld, int B, int HW, int C, vaw_stream stream);
int vaw_rowvec_sum(vaw_dtype dt, const void* dh, float* de, int64_t ld, int B, int HW, int C, float beta, vaw_stream stream);
int vaw_nchw_to_nhwc(vaw_dtype dt, const float* nchw, void* nhwc, int B, int C, int HW, vaw_stream stream);
int vaw_nhwc_to_nchw(vaw_dtype dt, const void* nhwc, float* nchw, int B, int C, int HW, vaw_stream stream);

/* ---------------------------------------------------------------------------
 * Optimizer side  (torch.optim.AdamW at main.py:354, ema() tools/trainer.py:12-18,
 * clip_grad_norm_ tools/trainer.py:60-62)
 * ------------------------------------------------------------------------- */

/* sumsq_out[0] = (accumulate ? sumsq_out[0] : 0) + sum g^2 over n elements; fixed-order two-stage reduction
 * through a workspace of vaw_sumsq_workspace_floats() f32. */
int64_t vaw_sumsq_workspace_floats(void);
int vaw_sumsq(const float* g, int64_t n, float* sumsq_out, int accumulate, float* workspace, vaw_stream stream);
/* One fused pass over flat f32 buffers: decoupled-weight-decay Adam, then EMA, then the bf16 shadow copy.
 *   g' = g * gscale, gscale = clip_max_norm>0 ? min(1, clip_max_norm/(sqrt(*sumsq)+1e-6)) : 1
 *   p *= 1 - lr*wd ; m += (g'-m)(1-b1) ; v = b2 v + (1-b2) g'^2
 *   p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
 *   ema = ema*decay + p*(1-decay)   (if ema != NULL)
 *   shadow = bf16(p)                (if shadow != NULL)
 *   g = 0                           (if zero_grad)
 * bc1 = 1-b1^step, bc2 = 1-b2^step are computed on the host. */
int vaw_adamw_ema_step(float* p, float* g, float* m, float* v, float* ema, void* shadow_bf16, int64_t n, float lr,
                       float beta1, float beta2, float eps, float weight_decay, float bc1, float bc2,
                       float ema_decay, const float* sumsq, float clip_max_norm, int zero_grad, vaw_stream stream);
/* Same update with the step-dependent scalars read from DEVICE memory -- hyper = f32[3] {lr, bc1, bc2} -- so the launch
 * carries no per-step host value and a captured hipGraph of the whole training step can be replayed while the LR
 * schedule (LambdaLR, utils.py:75-90) and the bias corrections advance (the host refreshes hyper before each replay). */
int vaw_adamw_ema_step_dev(float* p, float* g, float* m, float* v, float* ema, void* shadow_bf16, int64_t n,
                           const float* hyper, float beta1, float beta2, float eps, float weight_decay, float ema_decay,
                           const float* sumsq, float clip_max_norm, int zero_grad, vaw_stream stream);
/* ema = ema*decay + src*(1-decay) over n f32 (buffers that are not optimizer-owned, e.g. frozen pos_embed) */
int vaw_ema_update(float* ema, const float* src, int64_t n, float decay, vaw_stream stream);
/* dst(bf16) = src(f32) */
int vaw_cast_bf16(const float* src, void* dst, int64_t n, vaw_stream stream);
/* dst[i] = scale * float(src[i]), src bf16: the way back of a gradient bucket that was all-reduced in bf16 over xGMI
 * (the reference's DDP reduces f32 buckets, main.py:347-348; bf16 halves the bytes on the wire); scale = 1/world when the
 * collective summed.  src and dst 16-byte aligned. */
int vaw_uncast_bf16(const void* src, float* dst, int64_t n, float scale, vaw_stream stream);

/* ---- gradient-bucket collectives straight on RCCL ------------------------------------------------
 * Replaces the bucket all-reduce torch DDP does for the reference (main.py:347; process group set up in tools/dist_util.py:55).
 * One communicator per process (= per GPU) with a side HIP stream of its own; RCCL is opened with dlopen at the first call
 * (VAW_ERR_UNSUPPORTED when the box has none), so the rest of the library never depends on it.
 *   vaw_comm_unique_id(out)           rank 0: 128 opaque bytes to hand to every rank by any host channel
 *   vaw_comm_init(id, rank, world)    every rank (collective); vaw_comm_world() = 0 before it; vaw_comm_destroy() ends it
 *   vaw_allreduce_bucket_start        buf[0..count) <- MEAN over ranks, in place, ordered behind everything enqueued on `stream`
 *   vaw_reduce_scatter_bucket_start   rank r's chunk (count/world elements at buf + r*chunk) <- mean of that chunk (ZeRO-1 form)
 *   vaw_allgather_bucket_start        every rank's chunk of buf -> all ranks, in place
 *   vaw_allreduce_bucket_wait         `stream` waits for every collective started so far; the host never blocks
 * dt = VAW_F32 or VAW_BF16 (the wire type).  Not capturable into a hipGraph. */
int vaw_comm_unique_id(void* out128);
int vaw_comm_init(const void* id128, int rank, int world);
int vaw_comm_world(void);
int vaw_comm_destroy(void);
int vaw_allreduce_bucket_start(void* buf, int64_t count, vaw_dtype dt, vaw_stream stream);
int vaw_reduce_scatter_bucket_start(void* buf, int64_t count, vaw_dtype dt, vaw_stream stream);
int vaw_allgather_bucket_start(void* buf, int64_t count, vaw_dtype dt, vaw_stream stream);
int vaw_allreduce_bucket_wait(vaw_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* VAW_HIP_H */
