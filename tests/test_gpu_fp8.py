"""fp8 (OCP e4m3fn / e5m2) GEMM path of BASELINE.json config 5 through the C ABI: vaw_fp8_quantize, vaw_gemm_fp8 and
vaw_wgrad_grouped(dt = VAW_FP8).

  * the quantiser is byte-exact against torch's float8_e4m3fn cast of the same scaled values (round to nearest even) and its
    transposed copy is the transpose;
  * the GEMMs are checked with small integers (exact in e4m3, exact in the f32 accumulator) and power-of-two scales, so operand
    layout, block-scale encoding, tile edges, grouping, K split and fixup must all be right bit for bit;
  * the fused epilogues are checked against float64 arithmetic on the dequantised operands.
There is no reference fp8 code to pin against (the reference trains this model under bf16 autocast); the end-to-end drift of the fp8
training step against the reference's f32 fixture is asserted in test_gpu_bigcfg.py."""
import pytest
import torch

pytestmark = pytest.mark.gpu

import vaw_amd
from vaw_amd import ops
from vaw_amd._lib import BF8, FP8, ptr

DEV = "cuda"
E4M3, E5M2 = torch.float8_e4m3fn, torch.float8_e5m2
FMT = {"e4m3": (FP8, E4M3, 448.0), "e5m2": (BF8, E5M2, 57344.0)}


def _bytes(t, tdt=E4M3):
    """exactly representable float values -> fp8 bytes on the GPU"""
    assert torch.equal(t.to(tdt).float(), t)
    return t.to(tdt).view(torch.uint8).to(DEV)


@pytest.mark.parametrize("fmt", ["e4m3", "e5m2"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("R,C", [(256, 128), (200, 264), (1000, 68), (4, 4), (64, 2048)])
def test_quantize_bytes_and_transpose(dtype, R, C, fmt):
    code, E, FMAX = FMT[fmt]
    g = torch.Generator().manual_seed(R + C)
    x = (torch.randn(R, C, generator=g) * torch.logspace(-3, 1, C)[None, :]).to(dtype)     # wide dynamic range: subnormals, zeros
    x[0, 0] = 0.0
    f = ops.Fp8(R, C, torch.device(DEV), fmt=code).quantize(x.to(DEV))
    amax = x.float().abs().max()
    scale = amax / torch.tensor(FMAX)
    inv = torch.tensor(1.0) / scale
    ref = (x.float() * inv).to(E).view(torch.uint8)
    assert float(f.scale.cpu()) == float(scale)
    assert torch.equal(f.q.cpu(), ref)
    assert torch.equal(f.qt.cpu(), ref.t())
    # transposed copy only
    f2 = ops.Fp8(R, C, torch.device(DEV), plain=False, fmt=code).quantize(x.to(DEV))
    assert torch.equal(f2.qt.cpu(), ref.t())
    z = ops.Fp8(R, C, torch.device(DEV), fmt=code).quantize(torch.zeros(R, C, device=DEV, dtype=dtype))
    assert float(z.scale.cpu()) == 1.0 and int(z.q.max()) == 0
    # a row count off the 4-grid: no transposed copy (its row stride would not be a multiple of 4), plain copy still exact
    f3 = ops.Fp8(R - 1, C, torch.device(DEV), transposed=False, fmt=code).quantize(x[1:].contiguous().to(DEV))
    x3 = x[1:].float()
    s3 = x3.abs().max() / torch.tensor(FMAX)
    assert torch.equal(f3.q.cpu(), (x3 * (torch.tensor(1.0) / s3)).to(E).view(torch.uint8))


@pytest.mark.parametrize("R,C", [(520, 328), (576, 384), (640, 384)])        # generic kernel | the 64 x 128 | the 128 x 128 tiled kernel (training shapes)
@pytest.mark.parametrize("fmt", ["e4m3", "e5m2"])
def test_quantize_delayed_scaling_state(fmt, R, C):
    """vaw_fp8_quantize_delayed + vaw_fp8_scale_update: bytes = cast(clamp(x / scale)) with the scale in the state, the state's
    running max = max |x| exactly (atomic max on the bits), and the update turns it into scale = amax * margin / FMAX."""
    code, E, FMAX = FMT[fmt]
    dev = torch.device(DEV)
    g = torch.Generator().manual_seed(9)
    st = ops.fp8_states([code, code], dev, margin=2.0)
    f = ops.Fp8(R, C, dev, fmt=code, state=st[1])
    x1 = torch.randn(R, C, generator=g).bfloat16()
    f.quantize(x1.to(DEV))                                   # first step: just-in-time scale
    s1 = x1.float().abs().max() / torch.tensor(FMAX)
    # ... which also leaves this pass's max |x| in the running-max slot, so that a scale update right after a just-in-time
    # pass applies the margin like every later one (the first delayed step then has headroom)
    assert float(st[1, 0]) == float(s1) and float(st[1, 1]) == float(s1 * torch.tensor(FMAX))
    x2 = (torch.randn(R, C, generator=g) * 3).bfloat16()     # larger than the scale covers: saturates
    f.quantize(x2.to(DEV), delayed=True)
    ref = (x2.float() * (torch.tensor(1.0) / s1)).clamp(-FMAX, FMAX).to(E).view(torch.uint8)
    assert torch.equal(f.q.cpu(), ref) and torch.equal(f.qt.cpu(), ref.t())
    assert float(st[1, 1]) == float(x2.float().abs().max())
    ops.fp8_scale_update(st)
    assert float(st[1, 0]) == float(x2.float().abs().max() / torch.tensor(FMAX / 2.0)) and float(st[1, 1]) == 0.0
    assert float(st[0, 0]) == 1.0                            # a state nobody quantised with keeps its scale
    f.quantize(x2.to(DEV), delayed=True)
    s2 = st[1, 0].cpu()
    assert torch.equal(f.q.cpu(), (x2.float() * (torch.tensor(1.0) / s2)).to(E).view(torch.uint8))


def _int_operands(M, N, K, seed):
    g = torch.Generator().manual_seed(seed)
    A = torch.randint(-4, 5, (M, K), generator=g).float()
    B = torch.randint(-4, 5, (N, K), generator=g).float()
    # make the k positions distinguishable (a k permutation applied to one operand only would change the result)
    A[:, ::7] *= 2
    B[:, 1::5] *= 0.5
    return A, B


@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (512, 768, 1152), (1000, 264, 384), (4096, 1152, 1152), (272, 192, 256),
                                   (256, 2304, 256)])
@pytest.mark.parametrize("afmt", ["e4m3", "e5m2"])
def test_gemm_fp8_exact_integers(M, N, K, afmt):
    acode, AE, _ = FMT[afmt]
    A, B = _int_operands(M, N, K, M + N + K)
    sa = torch.tensor([0.5], device=DEV)
    sb = torch.tensor([4.0], device=DEV)
    bias = torch.randint(-3, 4, (N,)).float()
    ref = 2.0 * (A.double() @ B.double().t()) + bias.double()
    Ad, Bd, bd = _bytes(A, AE), _bytes(B), bias.to(DEV)
    out = torch.empty(M, N, device=DEV, dtype=torch.float32)
    ops.gemm_fp8(M, N, K, ptr(Ad), K, ptr(sa), ptr(Bd), K, ptr(sb), ptr(out), N, a_format=acode, bias=ptr(bd), out_f32=True)
    assert torch.equal(out.cpu().double(), ref), (out.cpu().double() - ref).abs().max()
    # bf16 output + column sums of what was stored
    outb = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    cs = torch.zeros(N, device=DEV)
    ops.gemm_fp8(M, N, K, ptr(Ad), K, ptr(sa), ptr(Bd), K, ptr(sb), ptr(outb), N, a_format=acode, colsum_out=ptr(cs))
    refb = (2.0 * (A.double() @ B.double().t())).float().bfloat16()
    assert torch.equal(outb.cpu(), refb)
    torch.testing.assert_close(cs.cpu().double(), refb.double().sum(0), rtol=1e-6, atol=1e-3)


def test_gemm_fp8_epilogues():
    M, N, K, T = 1024, 768, 512, 256
    g = torch.Generator().manual_seed(5)
    dev = torch.device(DEV)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) * 0.05
    fx, fw = ops.Fp8(M, K, dev).quantize(x.to(DEV)), ops.Fp8(N, K, dev).quantize(w.to(DEV))
    xd, wd = fx.dequant().cpu().double(), fw.dequant().cpu().double()
    bias = torch.randn(N, generator=g)
    bias_d = bias.to(DEV)          # (device operands live in named tensors: a temporary's memory is reused by the next .to())
    lin = xd @ wd.t() + bias.double()
    args = (M, N, K, ptr(fx.q), K, ptr(fx.scale), ptr(fw.q), K, ptr(fw.scale))
    # bias + aux_out + GELU(tanh)
    out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    aux = torch.empty_like(out)
    ops.gemm_fp8(*args, ptr(out), N, bias=ptr(bias_d), act=1, aux_out=ptr(aux))
    torch.testing.assert_close(aux.cpu().double(), lin, rtol=1e-2, atol=1e-2)
    torch.testing.assert_close(out.cpu().double(), torch.nn.functional.gelu(aux.cpu().double(), approximate="tanh"), rtol=1e-2, atol=1e-2)
    # GELU'(aux_in) + column sums
    pre = torch.randn(M, N, generator=g).bfloat16()
    cs = torch.zeros(N, device=DEV)
    pre_d = pre.to(DEV)
    ops.gemm_fp8(*args, ptr(out), N, act=2, aux_in=ptr(pre_d), colsum_out=ptr(cs))
    p = pre.double().requires_grad_(True)
    torch.nn.functional.gelu(p, approximate="tanh").sum().backward()
    torch.testing.assert_close(out.cpu().double(), (xd @ wd.t()) * p.grad, rtol=1e-2, atol=1e-2)
    torch.testing.assert_close(cs.cpu().double(), out.cpu().double().sum(0), rtol=1e-5, atol=1e-2)
    # bias + aux_out + gate + f32 residual
    gate = torch.randn(M // T, N, generator=g)
    resid = torch.randn(M, N, generator=g)
    outf = torch.empty(M, N, device=DEV, dtype=torch.float32)
    gate_d, resid_d = gate.to(DEV), resid.to(DEV)
    ops.gemm_fp8(*args, ptr(outf), N, bias=ptr(bias_d), aux_out=ptr(aux), gate=ptr(gate_d), gate_ld=N,
                 resid=ptr(resid_d), rows_per_batch=T, out_f32=True)
    want = aux.cpu().double() * gate.double().repeat_interleave(T, 0) + resid.double()
    torch.testing.assert_close(aux.cpu().double(), lin, rtol=1e-2, atol=1e-2)
    torch.testing.assert_close(outf.cpu().double(), want, rtol=1e-5, atol=1e-5)
    # an epilogue without an fp8 kernel is refused, not approximated
    with pytest.raises(vaw_amd.VawError):
        ops.gemm_fp8(*args, ptr(outf), N, act=1, out_f32=True)


@pytest.mark.parametrize("dyfmt", ["e5m2", "e4m3"])
@pytest.mark.parametrize("case", ["few_tiles_split", "full_rounds_plus_split", "n192", "edges_accumulate"])
def test_wgrad_grouped_fp8_exact_integers(case, dyfmt):
    """dW_p (+)= alpha_p * s_dy * s_x * dy_p^T x_p from the transposed e4m3 copies dy^T [M][K], x^T [N][K]."""
    code, DE, _ = FMT[dyfmt]
    g = torch.Generator().manual_seed(13)
    shapes, K, beta = {
        "few_tiles_split": ([(768, 768), (2304, 768), (256, 512)], 1024, 0.0),
        "full_rounds_plus_split": ([(512, 512)] * 70, 2560, 0.0),
        "n192": ([(384, 1152), (1152, 384), (256, 192)], 512, 0.0),
        "edges_accumulate": ([(200, 264), (520, 72), (136, 1000)], 384, 1.0),
    }[case]
    probs, keep, refs = [], [], []
    for i, (M, N) in enumerate(shapes):
        dyT = torch.randint(-3, 4, (M, K), generator=g).float()
        xT = torch.randint(-3, 4, (N, K), generator=g).float()
        dyT[:, ::3] *= 2
        dw0 = torch.randint(-5, 6, (M, N), generator=g).float()
        s_dy = torch.tensor([2.0 if i % 2 else 0.25], device=DEV)
        s_x = torch.tensor([0.5], device=DEV)
        alpha = 2.0 if i == 1 else 0.0          # 0 = 1
        dyd, xd, dwd = _bytes(dyT, DE), _bytes(xT), dw0.to(DEV).clone()
        keep += [dyd, xd, dwd, s_dy, s_x]
        probs.append((ptr(dyd), ptr(xd), ptr(dwd), M, N, K, K, N, alpha, 0, ptr(s_dy), ptr(s_x)))
        f = (alpha or 1.0) * float(s_dy) * float(s_x)
        refs.append((dwd, beta * dw0.double() + f * (dyT.double() @ xT.double().t())))
    grp = ops.WgradGroup(probs, K, torch.device(DEV))
    grp.launch(code, beta)
    for i, (got, ref) in enumerate(refs):
        assert torch.equal(got.cpu().double(), ref), (case, i, shapes[i], (got.cpu().double() - ref).abs().max())
    if beta == 0.0:
        grp.launch(code, 0.0)
        for got, ref in refs:
            assert torch.equal(got.cpu().double(), ref)


@pytest.mark.parametrize("fmt", ["e4m3", "e5m2"])
def test_row_kernels_with_fp8_output_match_the_quantiser(fmt):
    """vaw_ln_modulate_fwd_fp8 / vaw_gate_bwd_fp8 / vaw_fp8_transpose: the bytes, the transposed copy, the running max and the
    side outputs (mean / rstd, dgate, bias-gradient partials) must equal what the bf16 kernels followed by
    vaw_fp8_quantize_delayed produce with the same scale."""
    code, E, FMAX = FMT[fmt]
    dev = torch.device(DEV)
    B, T, D = 4, 64, 256
    M = B * T
    g = torch.Generator().manual_seed(21)
    x = torch.randn(M, D, generator=g).to(DEV)
    mod = (torch.randn(B, 3 * D, generator=g) * 0.3).to(DEV)
    scale0 = 0.01
    # --- LayerNorm + modulate forward
    st = ops.fp8_states([code, code], dev)
    st[:, 0] = scale0
    ref_bf16 = torch.empty(M, D, device=DEV, dtype=torch.bfloat16)
    mean1, rstd1 = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    ops.ln_modulate_fwd(vaw_amd._lib.BF16, ptr(x), ptr(mod), ptr(mod) + 4 * D, 3 * D, ptr(ref_bf16), ptr(mean1), ptr(rstd1), B, T, D)
    fa = ops.Fp8(M, D, dev, fmt=code, state=st[0]).quantize(ref_bf16, delayed=True)
    fb = ops.Fp8(M, D, dev, fmt=code, state=st[1])
    mean2, rstd2 = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    ops.ln_modulate_fwd_fp8(ptr(x), ptr(mod), ptr(mod) + 4 * D, 3 * D, fb, ptr(mean2), ptr(rstd2), B, T, D)
    fb.transpose_from_q()
    assert torch.equal(fb.q, fa.q) and torch.equal(fb.qt, fa.qt) and torch.equal(fb.qt, fb.q.t())
    assert float(st[1, 1]) == float(st[0, 1]) == float(ref_bf16.float().abs().max())
    assert torch.equal(mean1, mean2) and torch.equal(rstd1, rstd2)
    # --- gate backward
    st = ops.fp8_states([code, code], dev)
    st[:, 0] = scale0
    dres = torch.randn(M, D, generator=g).to(DEV)
    y = torch.randn(M, D, generator=g).to(DEV).bfloat16()
    dy = torch.empty(M, D, device=DEV, dtype=torch.bfloat16)
    dg1, cp1 = torch.empty(B, 3 * D, device=DEV), torch.empty(B, D, device=DEV)
    ops.gate_bwd(vaw_amd._lib.BF16, ptr(dres), ptr(y), ptr(mod) + 4 * 2 * D, 3 * D, ptr(dy), ptr(dg1), 3 * D, B, T, D, ptr(cp1))
    fa = ops.Fp8(M, D, dev, fmt=code, state=st[0]).quantize(dy, delayed=True)
    fb = ops.Fp8(M, D, dev, fmt=code, state=st[1])
    dg2, cp2 = torch.empty(B, 3 * D, device=DEV), torch.empty(B, D, device=DEV)
    ops.gate_bwd_fp8(ptr(dres), ptr(y), ptr(mod) + 4 * 2 * D, 3 * D, fb, ptr(dg2), 3 * D, B, T, D, ptr(cp2))
    fb.transpose_from_q()
    assert torch.equal(fb.q, fa.q) and torch.equal(fb.qt, fa.qt)
    assert float(st[1, 1]) == float(st[0, 1])
    assert torch.equal(dg1[:, :D], dg2[:, :D]) and torch.equal(cp1, cp2)


@pytest.mark.parametrize("R,C", [(192, 128), (256, 384), (1024, 1152), (64, 256)])
def test_fp8_transpose_kernels(R, C):
    """vaw_fp8_transpose: qt[c][r] = q[r][c] for any bytes; R % 128 == 0 takes the 128 x 128-tile kernel (16 bytes per lane in and
    out), other multiples of 64 the 64-row one."""
    g = torch.Generator().manual_seed(R + C)
    q = torch.randint(0, 256, (R, C), generator=g, dtype=torch.uint8).to(DEV)
    qt = torch.zeros(C, R, dtype=torch.uint8, device=DEV)
    ops.check(vaw_amd._lib.lib().vaw_fp8_transpose(ptr(q), R, C, C, ptr(qt), R, vaw_amd._lib.stream_ptr()), "vaw_fp8_transpose")
    assert torch.equal(qt, q.t())


@pytest.mark.parametrize("fmt", ["e4m3", "e5m2"])
@pytest.mark.parametrize("B,T,D", [(4, 64, 256), (2, 128, 1152)])
def test_ln_bwd_gate_fp8_matches_the_pair(fmt, B, T, D):
    """vaw_ln_modulate_bwd_gate_fp8 = vaw_ln_modulate_bwd followed by vaw_gate_bwd_fp8 on its dx: bytes, running max, dx and every
    per-sample sum bitwise equal (D = 1152: the 512-thread variant with the rows of a sample cut into chunks)."""
    code, E, FMAX = FMT[fmt]
    dev = torch.device(DEV)
    M = B * T
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(M, D, generator=g) * 2 + 0.5).to(DEV)
    mod = (torch.randn(B, 6 * D, generator=g) * 0.5).to(DEV)
    dout = torch.randn(M, D, generator=g).bfloat16().to(DEV)
    dres = torch.randn(M, D, generator=g).to(DEV)
    y = torch.randn(M, D, generator=g).bfloat16().to(DEV)
    mean, rstd = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    out = torch.empty(M, D, device=DEV, dtype=torch.bfloat16)
    BF = vaw_amd._lib.BF16
    ops.ln_modulate_fwd(BF, ptr(x), ptr(mod) + 4 * 3 * D, ptr(mod) + 4 * 4 * D, 6 * D, ptr(out), ptr(mean), ptr(rstd), B, T, D)

    def run(fused):
        st = ops.fp8_states([code], dev)
        st[:, 0] = 0.02
        f = ops.Fp8(M, D, dev, fmt=code, state=st[0])
        dmod = torch.zeros(B, 6 * D, device=DEV)
        dx, part = torch.empty(M, D, device=DEV), torch.empty(B, D, device=DEV)
        a = (ptr(dout), ptr(x), ptr(mean), ptr(rstd), ptr(mod) + 4 * 4 * D, 6 * D, ptr(dres), ptr(dx), ptr(dmod) + 4 * 3 * D,
             ptr(dmod) + 4 * 4 * D, 6 * D)
        if fused:       # the C entry point itself (ops.ln_modulate_bwd_gate_fp8 sends rows wider than 768 to the pair)
            ws = ops._row_ws(B, T, D)
            ops.check(vaw_amd._lib.lib().vaw_ln_modulate_bwd_gate_fp8(*a, ptr(y), ptr(mod) + 4 * 5 * D, f.epilogue_target(2), f.state.data_ptr(),
                                                                     f.fmt, ptr(dmod) + 4 * 5 * D, ptr(part), B, T, D, ws.data_ptr(), ws.numel(),
                                                                     vaw_amd._lib.stream_ptr()), "vaw_ln_modulate_bwd_gate_fp8")
        else:
            ops.ln_modulate_bwd(BF, *a, B, T, D)
            ops.gate_bwd_fp8(ptr(dx), ptr(y), ptr(mod) + 4 * 5 * D, 6 * D, f, ptr(dmod) + 4 * 5 * D, 6 * D, B, T, D, ptr(part))
        f.transpose_from_q()
        torch.cuda.synchronize()
        return dx, dmod, part, f.qt.clone(), st.clone()
    for u, v in zip(run(False), run(True)):
        assert torch.equal(u, v)


def test_fp8_quantize_batched_matches_per_tensor_calls():
    """vaw_fp8_quantize_delayed_batched (all Linear weights of a step in one launch) == one vaw_fp8_quantize_delayed per tensor:
    bytes, transposed copies and running maxima."""
    dev = torch.device(DEV)
    shapes = [(1152, 1152), (3456, 1152), (200, 72), (64, 4608)]
    g = torch.Generator().manual_seed(4)
    srcs = [(torch.randn(R, C, generator=g) * (0.02 * (i + 1))).to(DEV) for i, (R, C) in enumerate(shapes)]
    st_a, st_b = ops.fp8_states([FP8] * len(shapes), dev), ops.fp8_states([FP8] * len(shapes), dev)
    st_a[:, 0] = st_b[:, 0] = 1e-4
    fa = [ops.Fp8(R, C, dev, state=st_a[i]) for i, (R, C) in enumerate(shapes)]
    fb = [ops.Fp8(R, C, dev, state=st_b[i]) for i, (R, C) in enumerate(shapes)]
    for f, x in zip(fa, srcs):
        f.quantize(x, delayed=True)
    grp = ops.Fp8QuantGroup([(ptr(x), f) for x, f in zip(srcs, fb)], dev)
    grp.launch()
    torch.cuda.synchronize()
    for a, b in zip(fa, fb):
        assert torch.equal(a.q, b.q) and torch.equal(a.qt, b.qt)
    assert torch.equal(st_a, st_b)
