"""Tensor-level wrappers over the C ABI (include/vaw_hip.h).  Each takes torch CUDA tensors, checks what the
kernel assumes about them (shape, dtype, contiguity) on the host, and launches on the current stream."""
import ctypes as C

import torch

from . import _lib as L
from ._lib import BF16, F32, AttnDesc, Epilogue, check, need_cuda, ptr, stream_ptr


def dt_of(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise L.VawError(f"unsupported activation dtype {t.dtype}")


def _f32c(*ts):
    for t in ts:
        if t is not None and (t.dtype != torch.float32 or not t.is_contiguous()):
            raise L.VawError(f"expected a contiguous float32 tensor, got {t.dtype} contiguous={t.is_contiguous()}")


# ---- diffusion objective -----------------------------------------------------------------------
def qsample(x0, noise, t, tab_a, tab_s):
    need_cuda(x0, noise, t, tab_a, tab_s)
    _f32c(x0, noise, tab_a, tab_s)
    assert noise.shape == x0.shape and t.dtype == torch.int64 and t.shape == (x0.shape[0],)
    out = torch.empty_like(x0)
    B = x0.shape[0]
    check(L.lib().vaw_qsample_fwd(ptr(x0), ptr(noise), ptr(t), ptr(tab_a), ptr(tab_s), tab_a.numel(), ptr(out), B,
                                  x0.numel() // B, stream_ptr()), "vaw_qsample_fwd")
    return out


def mix_rows(x, y, ca, cb):
    need_cuda(x, y, ca, cb)
    _f32c(x, y, ca, cb)
    assert x.shape == y.shape and ca.shape == cb.shape == (x.shape[0],)
    out = torch.empty_like(x)
    B = x.shape[0]
    check(L.lib().vaw_mix_rows(ptr(x), ptr(y), ptr(ca), ptr(cb), ptr(out), B, x.numel() // B, stream_ptr()), "vaw_mix_rows")
    return out


class _WeightedMSE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model_out, x0, noise, ca, cb, w):
        need_cuda(model_out, x0, noise, ca, cb, w)
        model_out = model_out.contiguous()
        _f32c(model_out, x0, noise, ca, cb, w)
        B = x0.shape[0]
        assert model_out.shape == x0.shape == noise.shape and ca.shape == cb.shape == w.shape == (B,)
        mse = torch.empty(B, device=x0.device, dtype=torch.float32)
        check(L.lib().vaw_wmse_fwd(ptr(model_out), ptr(x0), ptr(noise), ptr(ca), ptr(cb), ptr(w), ptr(mse), B,
                                   x0.numel() // B, stream_ptr()), "vaw_wmse_fwd")
        ctx.save_for_backward(model_out, x0, noise, ca, cb, w)
        return mse

    @staticmethod
    def backward(ctx, gmse):
        model_out, x0, noise, ca, cb, w = ctx.saved_tensors
        gmse = gmse.contiguous().float()
        dout = torch.empty_like(model_out)
        B = x0.shape[0]
        check(L.lib().vaw_wmse_bwd(ptr(model_out), ptr(x0), ptr(noise), ptr(ca), ptr(cb), ptr(w), ptr(gmse), ptr(dout),
                                   B, x0.numel() // B, stream_ptr()), "vaw_wmse_bwd")
        return dout, None, None, None, None, None


def weighted_mse(model_out, x0, noise, ca, cb, w):
    """mse[b] = w[b] * mean((ca[b]*x0 + cb[b]*noise - model_out)^2); differentiable in model_out."""
    return _WeightedMSE.apply(model_out, x0, noise, ca, cb, w)


class _VbTerms(torch.autograd.Function):
    """vb[b] of reference _vb_terms_bpd (gaussian_diffusion.py:775-808), one fused pass; d/d(var values) and, when the
    mean prediction is not detached (pure KL losses), d/d(mean output)."""

    @staticmethod
    def forward(ctx, mean_out, var_out, x0, x_t, coef, mean_mode, var_mode, scale):
        need_cuda(mean_out, x0, x_t, coef)
        _f32c(mean_out, x0, x_t, coef)
        B = x0.shape[0]
        assert mean_out.shape == x0.shape == x_t.shape and coef.shape == (B, 8)
        if var_out is not None:
            need_cuda(var_out)
            _f32c(var_out)
            assert var_out.shape == x0.shape
        vb = torch.empty(B, device=x0.device, dtype=torch.float32)
        check(L.lib().vaw_vb_fwd(ptr(mean_out), ptr(var_out) if var_out is not None else None, ptr(x0), ptr(x_t), ptr(coef),
                                 mean_mode, var_mode, scale, ptr(vb), B, x0.numel() // B, stream_ptr()), "vaw_vb_fwd")
        ctx.save_for_backward(mean_out, var_out, x0, x_t, coef)
        ctx.modes = (mean_mode, var_mode, scale)
        return vb

    @staticmethod
    def backward(ctx, gvb):
        mean_out, var_out, x0, x_t, coef = ctx.saved_tensors
        mean_mode, var_mode, scale = ctx.modes
        gvb = gvb.contiguous().float()
        d_mean = torch.empty_like(mean_out) if ctx.needs_input_grad[0] else None
        d_var = torch.empty_like(var_out) if (var_out is not None and ctx.needs_input_grad[1]) else None
        B = x0.shape[0]
        if d_mean is not None or d_var is not None:
            check(L.lib().vaw_vb_bwd(ptr(mean_out), ptr(var_out) if var_out is not None else None, ptr(x0), ptr(x_t), ptr(coef),
                                     mean_mode, var_mode, scale, ptr(gvb), ptr(d_mean) if d_mean is not None else None,
                                     ptr(d_var) if d_var is not None else None, B, x0.numel() // B, stream_ptr()), "vaw_vb_bwd")
        return d_mean, d_var, None, None, None, None, None, None


def vb_terms(mean_out, var_out, x0, x_t, coef, mean_mode, var_mode, scale=1.0):
    """Per-sample variational-bound term in bits/dim; see vaw_vb_fwd in include/vaw_hip.h."""
    return _VbTerms.apply(mean_out.contiguous(), None if var_out is None else var_out.contiguous(), x0, x_t, coef,
                          int(mean_mode), int(var_mode), float(scale))


def sample_step(kind, mean_out, var_out, x, noise, coef, mean_mode, var_mode, clip_denoised, eta=0.0, want_all=False):
    """One reverse-process step (vaw_sample_step): kind 0 = p_mean_variance only, 1 = p_sample, 2 = ddim_sample.
    Returns {"sample", "pred_xstart"} (+ "mean", "log_variance" with want_all)."""
    need_cuda(mean_out, x, coef)
    mean_out, x = mean_out.contiguous().float(), x.contiguous().float()
    var_out = None if var_out is None else var_out.contiguous().float()
    noise = None if noise is None else noise.contiguous().float()
    coef = coef.contiguous()
    B = x.shape[0]
    assert mean_out.shape == x.shape and coef.shape == (B, 16) and coef.dtype == torch.float32
    res = {"pred_xstart": torch.empty_like(x)}
    if kind:
        res["sample"] = torch.empty_like(x)
    if want_all:
        res["mean"], res["log_variance"] = torch.empty_like(x), torch.empty_like(x)
    check(L.lib().vaw_sample_step(kind, ptr(mean_out), ptr(var_out), ptr(x), ptr(noise), ptr(coef), int(mean_mode), int(var_mode),
                                  1 if clip_denoised else 0, float(eta), ptr(res.get("sample")), ptr(res["pred_xstart"]),
                                  ptr(res.get("mean")), ptr(res.get("log_variance")), B, x.numel() // B, stream_ptr()),
          "vaw_sample_step")
    return res


# ---- dense -------------------------------------------------------------------------------------
def gemm(dt, a_kmajor, b_kmajor, M, N, K, A, lda, B, ldb, Cp, ldc, *, bias=None, act=0, aux_in=None, aux_out=None,
         gate=None, gate_ld=0, resid=None, rowadd=None, rows_per_batch=0, alpha=1.0, beta=0.0, out_f32=False,
         colsum_out=None, colsum_beta=0.0, resid_is_act=False, rowsum_a_out=None, rowsum_a_beta=0.0, colsum_partial=None):
    """Raw-pointer GEMM; A, B, Cp, bias... are integers (device addresses).  See vaw_gemm in the header.
    colsum_partial = ColsumPartial: the column sums of C stay as partial rows in its buffer (folded later, many at once)."""
    e = Epilogue(bias or None, act, aux_in or None, aux_out or None, gate or None, gate_ld, resid or None,
                 rowadd or None, rows_per_batch, alpha, beta, 1 if out_f32 else 0, colsum_out or None, colsum_beta,
                 1 if resid_is_act else 0, rowsum_a_out or None, rowsum_a_beta)
    if colsum_partial is not None:
        colsum_partial.rows.value = colsum_partial.buf.shape[0]           # in: capacity; out: rows written
        e.colsum_partial_out, e.colsum_rows_out = colsum_partial.buf.data_ptr(), C.pointer(colsum_partial.rows)
    tr = gemm_trace
    if tr is not None:
        e0, e1 = tr.events()
        e0.record()
    ws_ptr, ws_n = 0, 0
    if colsum_out or colsum_partial is not None or rowsum_a_out or (beta_or_plain(bias, act, aux_out, gate, resid, rowadd) and K >= 2048):     # may run split-K
        ws = scratch_f32(torch.device("cuda", torch.cuda.current_device()), 0)
        ws_ptr, ws_n = ws.data_ptr(), ws.numel()
    check(L.lib().vaw_gemm(dt, 1 if a_kmajor else 0, 1 if b_kmajor else 0, M, N, K, A, lda, B, ldb, Cp, ldc,
                           C.byref(e), ws_ptr, ws_n, stream_ptr()), "vaw_gemm")
    if tr is not None:
        e1.record()
        es = 2 if dt == BF16 else 4
        nb = es * (M * K + N * K) + M * N * ((4 if (out_f32 or dt == F32) else 2) + (es if aux_out else 0) + (es if aux_in else 0) +
                                             ((es if resid_is_act else 4) if resid else 0) + (4 if (beta and out_f32) else 0))
        tr.add(L.lib().vaw_gemm_uses_bf16_mfma(dt, M, N, K, A, lda, B, ldb), bool(a_kmajor), bool(b_kmajor), M, N, K, e0, e1, float(nb))


# ---- fp8 operands ------------------------------------------------------------------------------------
def fp8_quantize(src_dt, src, R, C_, ld, q, qt, scale, fmt=L.FP8, device=None):
    """Raw-pointer vaw_fp8_quantize: src [R][C] (row stride ld) -> q [R][C] bytes, qt [C][R] bytes (0 = not wanted), scale."""
    ws = scratch_f32(device or torch.device("cuda", torch.cuda.current_device()), 0)
    check(L.lib().vaw_fp8_quantize(src_dt, fmt, src, R, C_, ld, q, C_, qt or None, R, scale, ws.data_ptr(), ws.numel(), stream_ptr()),
          "vaw_fp8_quantize")


FP8_MAX = {L.FP8: 448.0, L.BF8: 57344.0}


def fp8_states(formats, device, margin=2.0):
    """[n][4] f32 scaling states {scale, running amax, FMAX / margin, 0} for delayed scaling (vaw_fp8_quantize_delayed /
    vaw_fp8_scale_update), one row per tensor format in `formats`."""
    st = torch.zeros(len(formats), 4)
    st[:, 0] = 1.0
    st[:, 2] = torch.tensor([FP8_MAX[f] / margin for f in formats])
    return st.to(device)


def fp8_scale_update(states):
    check(L.lib().vaw_fp8_scale_update(states.data_ptr(), states.shape[0], stream_ptr()), "vaw_fp8_scale_update")


class Fp8:
    """An fp8 copy of a 2-D operand: q [R][C] bytes, optionally qt [C][R] (the transposed copy the k-major-only fp8 GEMMs read
    for the other contraction), and the device scalar `scale` that turns the bytes back into values.  fmt: FP8 (e4m3) | BF8 (e5m2).
    plain=False keeps only the transposed copy (the row-major bytes go to the buffer `q_scratch` hands out).  state: a row of
    fp8_states() shared with the once-per-step scale update (delayed scaling); without one the object owns a private row."""

    def __init__(self, R, C_, device, transposed=True, plain=True, fmt=L.FP8, state=None):
        self.R, self.C, self.fmt, self.device = int(R), int(C_), fmt, device
        self.q = torch.empty(R, C_, device=device, dtype=torch.uint8) if plain else None
        self.qt = torch.empty(C_, R, device=device, dtype=torch.uint8) if transposed else None
        self.state = state if state is not None else fp8_states([fmt], device)[0]
        self.scale = self.state[0:1]

    def quantize(self, src, ld=None, src_dt=None, delayed=False):
        """src: f32 / bf16 tensor holding R x C values with row stride ld (default C), or a device address with src_dt.
        delayed=False: scale = amax / FMAX taken from src first (two passes); True: the scale already in the state."""
        if isinstance(src, torch.Tensor):
            need_cuda(src)
            src_dt, src = dt_of(src), src.data_ptr()
        q = self.q if self.q is not None else q_scratch(self.R * self.C, self.device)
        if delayed:
            check(L.lib().vaw_fp8_quantize_delayed(src_dt, self.fmt, src, self.R, self.C, ld or self.C, q.data_ptr(), self.C, ptr(self.qt),
                                                   self.R, self.state.data_ptr(), stream_ptr()), "vaw_fp8_quantize_delayed")
        else:
            fp8_quantize(src_dt, src, self.R, self.C, ld or self.C, q.data_ptr(), ptr(self.qt), self.scale.data_ptr(), self.fmt, self.device)
            # leave this pass's max |x| in the running-max slot too: the first vaw_fp8_scale_update after a just-in-time pass then
            # yields amax * margin / FMAX like every later one (otherwise the first delayed step would run without headroom)
            torch.mul(self.state[0:1], FP8_MAX[self.fmt], out=self.state[1:2])
        self.last_q = q.data_ptr()
        return self

    def epilogue_target(self, pool=1):
        """Row-major byte buffer a producing kernel may write this tensor's fp8 form into (gemm_fp8(out_fp8=self),
        ln_modulate_fwd_fp8, gate_bwd_fp8); finish with transpose_from_q().  Pools keep a GEMM's fp8 input (pool 0: quantiser
        outputs, pool 2: row-kernel outputs) apart from the fp8 output its epilogue writes (pool 1)."""
        q = self.q if self.q is not None else q_scratch(self.R * self.C, self.device, pool=pool)
        self.last_q = q.data_ptr()
        return self.last_q

    def transpose_from_q(self):
        check(L.lib().vaw_fp8_transpose(self.last_q, self.R, self.C, self.C, self.qt.data_ptr(), self.R, stream_ptr()), "vaw_fp8_transpose")
        return self

    def dequant(self):
        return self.q.view(torch.float8_e5m2 if self.fmt == L.BF8 else torch.float8_e4m3fn).float() * self.scale


class Fp8QuantJob(C.Structure):
    """vaw_fp8_quant_job of include/vaw_hip.h."""
    _fields_ = [("src", C.c_void_p), ("q", C.c_void_p), ("qt", C.c_void_p), ("state", C.c_void_p), ("R", C.c_int64), ("C", C.c_int64),
                ("ld", C.c_int64), ("ldq", C.c_int64), ("ldt", C.c_int64)]


class Fp8QuantGroup:
    """Delayed-scaling e4m3 quantisation of many f32 tensors (src address, Fp8 with q and qt) in ONE launch
    (vaw_fp8_quantize_delayed_batched); the addresses stay put between steps, so the device table is uploaded once."""

    def __init__(self, pairs, device):
        jobs = [Fp8QuantJob(src, f.q.data_ptr(), ptr(f.qt), f.state.data_ptr(), f.R, f.C, f.C, f.C, f.R) for src, f in pairs]
        self.n = len(jobs)
        self.table = (Fp8QuantJob * self.n)(*jobs)
        self.fp8s = [f for _, f in pairs]
        self.desc = torch.empty(L.lib().vaw_fp8_quantize_batched_desc_bytes(self.n), device=device, dtype=torch.uint8)
        self.uploaded = False

    def launch(self):
        check(L.lib().vaw_fp8_quantize_delayed_batched(self.n, C.cast(self.table, C.c_void_p), self.desc.data_ptr(),
                                                       0 if self.uploaded else 1, stream_ptr()), "vaw_fp8_quantize_delayed_batched")
        self.uploaded = True
        for f in self.fp8s:
            f.last_q = f.q.data_ptr()


_q_scratch = {}


def q_scratch(n, device, pool=0):
    """Grow-only byte scratch for row-major fp8 copies that only the very next GEMM reads (pool 0: quantiser outputs, pool 1:
    tensors a GEMM epilogue writes as fp8 while its input operand sits in pool 0)."""
    t = _q_scratch.get((device, pool))
    if t is None or t.numel() < n:
        t = _q_scratch[(device, pool)] = torch.empty(n, device=device, dtype=torch.uint8)
    return t


def gemm_fp8(M, N, K, A, lda, scale_a, B, ldb, scale_b, Cp, ldc, *, a_format=L.FP8, bias=None, act=0, aux_in=None, aux_out=None,
             gate=None, gate_ld=0, resid=None, rows_per_batch=0, alpha=1.0, out_f32=False, colsum_out=None, colsum_beta=0.0,
             out_fp8=None, colsum_partial=None):
    """Raw-pointer fp8 GEMM (vaw_gemm_fp8): A [M][K] bytes of a_format, B [N][K] e4m3 bytes; scale_a / scale_b device scalars.
    out_fp8 = an Fp8 whose delayed-scaling state is current: C (= its row-major bytes) is written as fp8 by the epilogue."""
    e = Epilogue(bias or None, act, aux_in or None, aux_out or None, gate or None, gate_ld, resid or None, None, rows_per_batch,
                 alpha, 0.0, 1 if out_f32 else 0, colsum_out or None, colsum_beta, 0, None, 0.0)
    if colsum_partial is not None:
        colsum_partial.rows.value = colsum_partial.buf.shape[0]           # in: capacity; out: rows written
        e.colsum_partial_out, e.colsum_rows_out = colsum_partial.buf.data_ptr(), C.pointer(colsum_partial.rows)
    tr = gemm_trace
    if tr is not None:
        e0, e1 = tr.events()
        e0.record()
    ws_ptr, ws_n = 0, 0
    if colsum_out:
        ws = scratch_f32(torch.device("cuda", torch.cuda.current_device()), 0)
        ws_ptr, ws_n = ws.data_ptr(), ws.numel()
    check(L.lib().vaw_gemm_fp8(a_format, M, N, K, A, lda, scale_a, B, ldb, scale_b, Cp, ldc, C.byref(e),
                               out_fp8.state.data_ptr() if out_fp8 is not None else None, out_fp8.fmt if out_fp8 is not None else 0,
                               ws_ptr, ws_n, stream_ptr()), "vaw_gemm_fp8")
    if tr is not None:
        e1.record()
        nb = M * K + N * K + M * N * ((4 if out_f32 else 1 if out_fp8 is not None else 2) + (2 if aux_out else 0) + (2 if aux_in else 0) +
                                      (4 if resid else 0))
        tr.add(2, True, a_format == L.FP8, M, N, K, e0, e1, float(nb))


class WgradProblem(C.Structure):
    """vaw_wgrad_problem of include/vaw_hip.h."""
    _fields_ = [("dy", C.c_void_p), ("x", C.c_void_p), ("dw", C.c_void_p), ("M", C.c_int64), ("N", C.c_int64),
                ("ld_dy", C.c_int64), ("ld_x", C.c_int64), ("ld_dw", C.c_int64), ("alpha", C.c_float), ("pad_", C.c_int),
                ("scale_dy", C.c_void_p), ("scale_x", C.c_void_p)]


class WgradGroup:
    """The weight gradients dW_p (+)= dy_p^T x_p of many Linear layers as ONE launch (vaw_wgrad_grouped).  `problems` is a list of
    (dy_addr, x_addr, dw_addr, M, N, ld_dy, ld_x, ld_dw); the addresses are workspace / flat-buffer addresses that stay put
    between steps, so the device copy of the table is uploaded once."""

    def __init__(self, problems, K, device):
        self.n, self.K, self.device = len(problems), int(K), device
        self.table = (WgradProblem * self.n)(*[WgradProblem(*p) for p in problems])
        self.flop = sum(2.0 * p[3] * p[4] * K for p in problems)
        self.desc = torch.empty(L.lib().vaw_wgrad_grouped_desc_bytes(self.n), device=device, dtype=torch.uint8)
        self.uploaded = False

    def launch(self, dt, beta):
        ws = scratch_f32(self.device, 0)
        tr = gemm_trace
        if tr is not None:
            e0, e1 = tr.events()
            e0.record()
        check(L.lib().vaw_wgrad_grouped(dt, self.n, C.cast(self.table, C.c_void_p), self.K, float(beta), self.desc.data_ptr(),
                                        0 if self.uploaded else 1, ws.data_ptr(), ws.numel(), stream_ptr()), "vaw_wgrad_grouped")
        self.uploaded = True
        if tr is not None:
            e1.record()
            es = 1 if dt in (L.FP8, L.BF8) else 2
            tr.add_flop(2 if dt in (L.FP8, L.BF8) else 1, False, False, self.flop, e0, e1,
                        float(sum(es * (p.M + p.N) * self.K + 4 * p.M * p.N * (2 if beta else 1) for p in self.table)))


def beta_or_plain(bias, act, aux_out, gate, resid, rowadd):
    """True when the epilogue is alpha/beta only: the launches that may run split-K."""
    return not (bias or act or aux_out or gate or resid or rowadd)


def conv3x3(dt, mode, act, act2, w, out, B, H, W, Ci, Co, *, bias=None, resid=None, resid_is_act=True, beta=0.0,
            colsum_out=None, colsum_beta=0.0, rowsum_a_out=None, rowsum_a_beta=0.0):
    """Implicit-GEMM conv3x3 (vaw_conv3x3).  Returns False -- nothing launched -- when the shape needs the explicit
    im2col + GEMM path.  mode 0 forward, 1 input gradient, 2 weight gradient (f32 out, beta accumulates)."""
    e = Epilogue(bias or None, 0, None, None, None, 0, resid or None, None, 0, 1.0, beta, 1 if mode == 2 else 0,
                 colsum_out or None, colsum_beta, 1 if resid_is_act else 0, rowsum_a_out or None, rowsum_a_beta)
    ws = scratch_f32(torch.device("cuda", torch.cuda.current_device()), 0)
    tr = gemm_trace
    if tr is not None:
        e0, e1 = tr.events()
        e0.record()
    rc = L.lib().vaw_conv3x3(dt, mode, act, act2 or None, w, out, B, H, W, Ci, Co, C.byref(e), ws.data_ptr(), ws.numel(), stream_ptr())
    if rc == -3:
        if dt == L.BF16:       # the f32 parity mode always takes the explicit path; in bf16 it is a performance cliff worth a word
            key = ("conv3x3", ("fwd", "dgrad", "wgrad")[mode], B, H, W, Ci, Co)
            fallbacks[key] = fallbacks.get(key, 0) + 1
            if fallbacks[key] == 1:
                import warnings
                warnings.warn(f"vaw_amd: conv3x3 {key[1]} [B={B}, {H}x{W}, Ci={Ci}, Co={Co}] is not covered by the implicit-GEMM kernels "
                              "(channel counts off the 64 / 8 grid?): falling back to im2col + GEMM; counts in vaw_amd.ops.fallbacks",
                              RuntimeWarning, stacklevel=3)
        return False
    check(rc, "vaw_conv3x3")
    if tr is not None:
        e1.record()
        M = B * H * W
        nb = 2 * M * (Ci + Co) + (2 if mode != 2 else 4 * (2 if beta else 1)) * 9 * Ci * Co + ((2 * M * Co) if (mode == 0 and resid) else 0)
        tr.add(1, mode != 2, mode == 0, *((M, Co, 9 * Ci) if mode == 0 else (M, Ci, 9 * Co) if mode == 1 else (Co, 9 * Ci, M)), e0, e1, float(nb), tag=f"conv{H}x{W}")
    return True


fallbacks = {}       # (op, variant, shape...) -> times a bf16 launch left the fast kernels for the explicit path


class GemmTrace:
    """Measurement aid (bench.py): brackets every GEMM launch with HIP events on the launch stream."""

    def __init__(self):
        self.rows = []

    def events(self):
        return torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def add(self, mfma, ak, bk, M, N, K, e0, e1, nbytes=0.0, tag="gemm"):
        self.rows.append((mfma, ak, bk, 2.0 * M * N * K, e0, e1, nbytes, (tag, int(M), int(N), int(K))))

    def add_flop(self, mfma, ak, bk, flop, e0, e1, nbytes=0.0):
        self.rows.append((mfma, ak, bk, flop, e0, e1, nbytes, None))

    @staticmethod
    def _name(mfma, ak, bk):
        return ("fp8_mfma" if mfma == 2 else "bf16_mfma" if mfma else "generic_f32mfma") + ("/fwd" if ak and bk else "/dgrad" if ak else "/wgrad")

    def by_shape(self):
        """-> [(variant, (M, N, K) or None, launches, ms, flop)] sorted by time, after a device synchronize (bench.py --shape-table)."""
        acc = {}
        for mfma, ak, bk, flop, e0, e1, nbytes, shape in self.rows:
            d = acc.setdefault((self._name(mfma, ak, bk), shape), [0, 0.0, 0.0])
            d[0] += 1
            d[1] += e0.elapsed_time(e1)
            d[2] += flop
        return sorted(((k[0], k[1], v[0], v[1], v[2]) for k, v in acc.items()), key=lambda r: -r[3])

    def summarize(self):
        """-> {variant: {launches, flop, ms}} after a device synchronize."""
        out = {}
        for mfma, ak, bk, flop, e0, e1, nbytes, _shape in self.rows:
            name = self._name(mfma, ak, bk)
            d = out.setdefault(name, {"launches": 0, "flop": 0.0, "ms": 0.0, "bytes": 0.0})
            d["launches"] += 1
            d["flop"] += flop
            d["bytes"] += nbytes          # algorithmic: every operand and result once
            d["ms"] += e0.elapsed_time(e1)
        return out


gemm_trace = None


def gemm_t(A, B, *, a_kmajor=True, b_kmajor=True, out_dtype=None, bias=None, act=0, aux_in=None, want_aux=False,
           gate=None, resid=None, rowadd=None, rows_per_batch=0, alpha=1.0, beta=0.0, out=None, colsum_out=None,
           colsum_beta=0.0, rowsum_a_out=None, rowsum_a_beta=0.0):
    """Tensor-level GEMM for tests and small call sites: A, B 2-D contiguous, same dtype."""
    need_cuda(A, B)
    assert A.dim() == 2 and B.dim() == 2 and A.is_contiguous() and B.is_contiguous() and A.dtype == B.dtype
    dt = dt_of(A)
    M, K = (A.shape if a_kmajor else A.shape[::-1])
    N, Kb = (B.shape if b_kmajor else B.shape[::-1])
    assert K == Kb, (A.shape, B.shape, a_kmajor, b_kmajor)
    out_dtype = out_dtype or A.dtype
    if out is None:
        out = torch.empty(M, N, device=A.device, dtype=out_dtype)
    aux = torch.empty(M, N, device=A.device, dtype=A.dtype) if want_aux else None
    _f32c(bias, gate, resid, rowadd)
    gemm(dt, a_kmajor, b_kmajor, M, N, K, ptr(A), A.shape[1], ptr(B), B.shape[1], ptr(out), N, bias=ptr(bias), act=act,
         aux_in=ptr(aux_in), aux_out=ptr(aux), gate=ptr(gate), gate_ld=(gate.shape[-1] if gate is not None else 0),
         resid=ptr(resid), rowadd=ptr(rowadd), rows_per_batch=rows_per_batch, alpha=alpha, beta=beta,
         out_f32=(out.dtype == torch.float32), colsum_out=ptr(colsum_out), colsum_beta=colsum_beta,
         rowsum_a_out=ptr(rowsum_a_out), rowsum_a_beta=rowsum_a_beta)
    return (out, aux) if want_aux else out


_scratch = {}


def scratch_f32(device, n):
    """Grow-only f32 scratch per device for the fixed-order reductions (column sums, gradient norm, split-K
    slabs).  256 MiB to start with: 64 M floats hold 8 slabs of any weight gradient of the configs up to DiT-XL / ADM."""
    t = _scratch.get(device)
    if t is None or t.numel() < n:
        t = _scratch[device] = torch.empty(max(n, 1 << 26), device=device, dtype=torch.float32)
    return t


def colsum(dt, X, M, N, ldx, out, beta=0.0, device=None):
    need = L.lib().vaw_colsum_workspace_floats(M, N)
    ws = scratch_f32(device or torch.device("cuda", torch.cuda.current_device()), need)
    check(L.lib().vaw_colsum(dt, X, M, N, ldx, out, beta, ptr(ws), ws.numel(), stream_ptr()), "vaw_colsum")


# ---- DiT pieces (raw pointers; shapes are checked by the caller that owns the buffers) ---------------
def ln_modulate_fwd(dt, x, shift, scale, mod_ld, out, mean, rstd, B, T, D, eps=1e-6):
    check(L.lib().vaw_ln_modulate_fwd(dt, x, shift, scale, mod_ld, out, mean, rstd, B, T, D, eps, stream_ptr()),
          "vaw_ln_modulate_fwd")


def ln_modulate_fwd_fp8(x, shift, scale, mod_ld, f8, mean, rstd, B, T, D, eps=1e-6, pool=2):
    """ln_modulate_fwd whose output goes straight into the Fp8 `f8` (row-major bytes + running max; delayed scaling state current)."""
    q = f8.epilogue_target(pool)
    check(L.lib().vaw_ln_modulate_fwd_fp8(x, shift, scale, mod_ld, q, f8.state.data_ptr(), f8.fmt, mean, rstd, B, T, D, eps, stream_ptr()),
          "vaw_ln_modulate_fwd_fp8")


def gate_bwd_fp8(dres, y, gate, mod_ld, f8, dgate, dmod_ld, B, T, D, dy_colpart=0, pool=2):
    """gate_bwd whose dy goes straight into the Fp8 `f8`."""
    ws = _row_ws(B, T, D)
    q = f8.epilogue_target(pool)
    check(L.lib().vaw_gate_bwd_fp8(dres, y, gate, mod_ld, q, f8.state.data_ptr(), f8.fmt, dgate, dmod_ld, dy_colpart or None, B, T, D,
                                   ws.data_ptr(), ws.numel(), stream_ptr()), "vaw_gate_bwd_fp8")


def _row_ws(B, T, D):
    return scratch_f32(torch.device("cuda", torch.cuda.current_device()), L.lib().vaw_row_bwd_workspace_floats(B, T, D))


def ln_modulate_bwd(dt, dout, x, mean, rstd, scale, mod_ld, dres_in, dx, dshift, dscale, dmod_ld, B, T, D):
    ws = _row_ws(B, T, D)
    check(L.lib().vaw_ln_modulate_bwd(dt, dout, x, mean, rstd, scale, mod_ld, dres_in or None, dx, dshift, dscale,
                                      dmod_ld, B, T, D, ws.data_ptr(), ws.numel(), stream_ptr()), "vaw_ln_modulate_bwd")


def ln_modulate_bwd_gate(dt, dout, x, mean, rstd, scale, mod_ld, dres_in, dx, dshift, dscale, dmod_ld, y_next, gate_next, dy_next,
                         dgate_next, B, T, D, dy_colpart=0):
    """ln_modulate_bwd + the gate_bwd that consumes its dx, one pass (vaw_ln_modulate_bwd_gate: row_bwd_fuse8_kernel for bf16 rows up
    to 1280 wide -- per-sample sums in LDS slabs, operand rows requested ahead).  Wider rows run the pair instead (bitwise the same
    results): the fused kernel's LDS slabs no longer fit and the register-accumulator form held 16 waves per CU at 3.3 TB/s."""
    if D > 1280:
        ln_modulate_bwd(dt, dout, x, mean, rstd, scale, mod_ld, dres_in, dx, dshift, dscale, dmod_ld, B, T, D)
        gate_bwd(dt, dx, y_next, gate_next, mod_ld, dy_next, dgate_next, dmod_ld, B, T, D, dy_colpart)
        return
    ws = _row_ws(B, T, D)
    check(L.lib().vaw_ln_modulate_bwd_gate(dt, dout, x, mean, rstd, scale, mod_ld, dres_in or None, dx, dshift, dscale, dmod_ld,
                                           y_next, gate_next, dy_next, dgate_next, dy_colpart or None, B, T, D, ws.data_ptr(),
                                           ws.numel(), stream_ptr()), "vaw_ln_modulate_bwd_gate")


def ln_modulate_bwd_gate_fp8(dout, x, mean, rstd, scale, mod_ld, dres_in, dx, dshift, dscale, dmod_ld, y_next, gate_next, f8,
                             dgate_next, B, T, D, dy_colpart=0, pool=2):
    """The fused pass with dy_next going straight into the Fp8 `f8` (as gate_bwd_fp8); wide rows: the pair, as above."""
    if D > 1280:
        ln_modulate_bwd(BF16, dout, x, mean, rstd, scale, mod_ld, dres_in, dx, dshift, dscale, dmod_ld, B, T, D)
        gate_bwd_fp8(dx, y_next, gate_next, mod_ld, f8, dgate_next, dmod_ld, B, T, D, dy_colpart, pool)
        return
    ws = _row_ws(B, T, D)
    q = f8.epilogue_target(pool)
    check(L.lib().vaw_ln_modulate_bwd_gate_fp8(dout, x, mean, rstd, scale, mod_ld, dres_in or None, dx, dshift, dscale, dmod_ld,
                                               y_next, gate_next, q, f8.state.data_ptr(), f8.fmt, dgate_next, dy_colpart or None,
                                               B, T, D, ws.data_ptr(), ws.numel(), stream_ptr()), "vaw_ln_modulate_bwd_gate_fp8")


class ColsumPartial:
    """Partial column sums [rows][N] f32 a dy-producing kernel leaves behind (rows: set by the kernel / known to the caller),
    folded later by a ReduceGroup."""

    def __init__(self, max_rows, N, device):
        self.buf = torch.empty(max_rows, N, device=device, dtype=torch.float32)
        self.N = int(N)
        self.rows = C.c_int64(0)


class ReduceJob(C.Structure):
    """vaw_reduce_job of include/vaw_hip.h."""
    _fields_ = [("partial", C.c_void_p), ("out", C.c_void_p), ("R", C.c_int64), ("N", C.c_int64)]


class ReduceGroup:
    """out_j = beta * out_j + column sums of partial_j for many (partial, R, N, out) jobs in ONE launch
    (vaw_reduce_rows_batched); addresses are workspace / flat-buffer addresses, so the device table is uploaded once."""

    def __init__(self, jobs, device):
        self.n = len(jobs)
        self.table = (ReduceJob * self.n)(*[ReduceJob(*j) for j in jobs])
        self.desc = torch.empty(L.lib().vaw_reduce_rows_batched_desc_bytes(self.n), device=device, dtype=torch.uint8)
        self.uploaded = False

    def launch(self, beta):
        check(L.lib().vaw_reduce_rows_batched(self.n, C.cast(self.table, C.c_void_p), float(beta), self.desc.data_ptr(),
                                              0 if self.uploaded else 1, stream_ptr()), "vaw_reduce_rows_batched")
        self.uploaded = True


def gate_bwd(dt, dres, y, gate, mod_ld, dy, dgate, dmod_ld, B, T, D, dy_colpart=0):
    ws = _row_ws(B, T, D)
    check(L.lib().vaw_gate_bwd(dt, dres, y, gate, mod_ld, dy, dgate, dmod_ld, dy_colpart or None, B, T, D, ws.data_ptr(),
                               ws.numel(), stream_ptr()), "vaw_gate_bwd")


def reduce_rows(partial, R, N, out, beta):
    check(L.lib().vaw_reduce_rows(partial, R, N, out, beta, stream_ptr()), "vaw_reduce_rows")


def attn_desc_token_major(B, H, T, hd):
    """qkv rows [B*T, 3*H*hd] (timm Attention); output rows [B*T, H*hd]."""
    return AttnDesc(B, H, T, hd, T * 3 * H * hd, hd, 3 * H * hd, 1, T * H * hd, hd, H * hd, 1, hd ** -0.5)


def attn_desc_channel_major(B, H, T, ch):
    """qkv [B, 3*H*ch, T] (UNet QKVAttention, new order); output [B, H*ch, T]."""
    return AttnDesc(B, H, T, ch, 3 * H * ch * T, ch * T, 1, T, H * ch * T, ch * T, 1, T, ch ** -0.5)


def attn_fwd(dt, desc, q, k, v, o, lse):
    check(L.lib().vaw_attn_fwd(dt, C.byref(desc), q, k, v, o, lse, stream_ptr()), "vaw_attn_fwd")


def attn_bwd(dt, desc, q, k, v, o, d_o, lse, delta, dq, dk, dv):
    check(L.lib().vaw_attn_bwd(dt, C.byref(desc), q, k, v, o, d_o, lse, delta, dq, dk, dv, stream_ptr()), "vaw_attn_bwd")


def attn_bwd_colsum(dt, desc, q, k, v, o, d_o, lse, delta, dq, dk, dv, partial):
    """attn_bwd that leaves the column sums of dq | dk | dv as partial rows in the ColsumPartial `partial` (the qkv bias
    gradient, folded later).  False -- nothing launched -- when the kernel path in use cannot (call attn_bwd + colsum then)."""
    partial.rows.value = partial.buf.shape[0]             # in: capacity; out: rows written
    rc = L.lib().vaw_attn_bwd_colsum(dt, C.byref(desc), q, k, v, o, d_o, lse, delta, dq, dk, dv, partial.buf.data_ptr(),
                                     C.byref(partial.rows), stream_ptr())
    if rc == -3:
        return False
    check(rc, "vaw_attn_bwd_colsum")
    return True


def timestep_embedding(t, dim, dtype=torch.float32, max_period=10000.0):
    """[cos | sin] embedding (tools/nn.py:103-121).  t: float tensor [B] on the GPU."""
    need_cuda(t)
    t = t.float().contiguous()
    out = torch.empty(t.shape[0], dim, device=t.device, dtype=dtype)
    check(L.lib().vaw_timestep_embedding(dt_of(out), ptr(t), ptr(out), t.shape[0], dim, max_period, stream_ptr()),
          "vaw_timestep_embedding")
    return out


# ---- optimizer ---------------------------------------------------------------------------------
def sumsq(g, out, accumulate=False):
    ws = scratch_f32(g.device, L.lib().vaw_sumsq_workspace_floats())
    check(L.lib().vaw_sumsq(ptr(g), g.numel(), ptr(out), 1 if accumulate else 0, ptr(ws), stream_ptr()), "vaw_sumsq")


def adamw_ema_step(p, g, m, v, ema, shadow, lr, beta1, beta2, eps, wd, step, ema_decay, sumsq_t, clip, zero_grad, hyper=None):
    """hyper: device f32[3] {lr, bc1, bc2} -- when given, the kernel reads the step-dependent scalars from it (graph replay)."""
    if hyper is not None:
        check(L.lib().vaw_adamw_ema_step_dev(ptr(p), ptr(g), ptr(m), ptr(v), ptr(ema), ptr(shadow), p.numel(), ptr(hyper), beta1,
                                             beta2, eps, wd, ema_decay, ptr(sumsq_t), clip or 0.0, 1 if zero_grad else 0,
                                             stream_ptr()), "vaw_adamw_ema_step_dev")
        return
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    check(L.lib().vaw_adamw_ema_step(ptr(p), ptr(g), ptr(m), ptr(v), ptr(ema), ptr(shadow), p.numel(), lr, beta1, beta2,
                                     eps, wd, bc1, bc2, ema_decay, ptr(sumsq_t), clip or 0.0, 1 if zero_grad else 0,
                                     stream_ptr()), "vaw_adamw_ema_step")


def ema_update(ema, src, decay):
    check(L.lib().vaw_ema_update(ptr(ema), ptr(src), ema.numel(), decay, stream_ptr()), "vaw_ema_update")


def cast_bf16(src, dst):
    check(L.lib().vaw_cast_bf16(ptr(src), ptr(dst), src.numel(), stream_ptr()), "vaw_cast_bf16")


def uncast_bf16(src, dst, scale=1.0):
    """dst (f32) = scale * src (bf16), on the current stream."""
    check(L.lib().vaw_uncast_bf16(ptr(src), ptr(dst), src.numel(), float(scale), stream_ptr()), "vaw_uncast_bf16")
