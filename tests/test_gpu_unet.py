"""Parity of the HIP UNet path: each UNet-side kernel against a float64 torch statement, the tiny UNets against
the reference's golden fixtures (forward, input gradient, every parameter gradient), and the cfg1 `Trainer`
trajectories of the reference.  Run on the MI355X box: pytest -m gpu."""
import copy
import random

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import Pbar, assert_fingerprints, base_args, load_json, load_pt, perturb_, synth_loader

pytestmark = pytest.mark.gpu

import vaw_amd
from vaw_amd import ops
from vaw_amd._lib import BF16, F32, lib, ptr, stream_ptr

DEV = "cuda"


def _rand(*shape, seed=0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def _nhwc(x):   # [B,C,H,W] -> [B*H*W, C]
    return x.permute(0, 2, 3, 1).reshape(-1, x.shape[1]).contiguous()


def _nchw(m, B, H, W):
    return m.reshape(B, H, W, -1).permute(0, 3, 1, 2).contiguous()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,C,H,film,silu", [(2, 64, 8, True, True), (3, 96, 5, False, True), (2, 32, 4, False, False), (1, 192, 16, True, True),
                                             (2, 64, 40, True, True)])
def test_groupnorm_film_silu_fwd_bwd(dtype, B, C, H, film, silu):
    tol = dict(rtol=2e-5, atol=2e-5) if dtype == torch.float32 else dict(rtol=3e-2, atol=3e-2)
    HW = H * H
    x = (_rand(B, C, H, H, seed=1) * 1.5 + 0.3).to(dtype)
    gamma, beta = _rand(C, seed=2) * 0.5 + 1, _rand(C, seed=3) * 0.2
    emb = _rand(B, 3 * C, seed=4) * 0.3
    dout = _rand(B, C, H, H, seed=5).to(dtype)
    dadd = _rand(B, C, H, H, seed=6).to(dtype)
    xr = x.double().requires_grad_(True)
    gr, br, er = gamma.double().requires_grad_(True), beta.double().requires_grad_(True), emb.double().requires_grad_(True)
    ref = F.group_norm(xr, 32, gr, br, eps=1e-5)
    if film:
        ref = ref * (1 + er[:, C:2 * C, None, None]) + er[:, 2 * C:, None, None]
    if silu:
        ref = F.silu(ref)
    (ref * dout.double()).sum().backward()
    dt = F32 if dtype == torch.float32 else BF16
    xd, gd, bd, ed = _nhwc(x).to(DEV), gamma.to(DEV), beta.to(DEV), emb.to(DEV)
    y = torch.empty(B * HW, C, device=DEV, dtype=dtype)
    mean, rstd = torch.empty(B * 32, device=DEV), torch.empty(B * 32, device=DEV)
    ws = torch.empty(lib().vaw_groupnorm_workspace_floats(B, HW, C), device=DEV)
    sc = ptr(ed) + 4 * C if film else None
    sh = ptr(ed) + 8 * C if film else None
    assert lib().vaw_groupnorm_fwd(dt, ptr(xd), ptr(gd), ptr(bd), sc, sh, 3 * C, int(silu), ptr(y), ptr(mean), ptr(rstd), B, HW, C, 32,
                                   1e-5, ptr(ws), stream_ptr()) == 0
    torch.testing.assert_close(_nchw(y.cpu().double(), B, H, H), ref.detach(), **tol)
    dod, dad = _nhwc(dout).to(DEV), _nhwc(dadd).to(DEV)
    dx = torch.empty_like(xd)
    dg, db = torch.ones(C, device=DEV), torch.ones(C, device=DEV)
    demb = torch.zeros(B, 3 * C, device=DEV)
    assert lib().vaw_groupnorm_bwd(dt, ptr(dod), ptr(xd), ptr(mean), ptr(rstd), ptr(gd), ptr(bd), sc, sh, 3 * C, int(silu), ptr(dad),
                                   ptr(dx), ptr(dg), ptr(db), 1.0, (ptr(demb) + 4 * C) if film else None,
                                   (ptr(demb) + 8 * C) if film else None, 3 * C, B, HW, C, 32, ptr(ws), stream_ptr()) == 0
    torch.testing.assert_close(_nchw(dx.cpu().double(), B, H, H), xr.grad + dadd.double(), **tol)
    rt = tol["rtol"]
    torch.testing.assert_close(dg.cpu().double(), 1 + gr.grad, rtol=rt, atol=tol["atol"] * 10)   # grad_beta=1 accumulates
    torch.testing.assert_close(db.cpu().double(), 1 + br.grad, rtol=rt, atol=tol["atol"] * 10)
    if film:
        torch.testing.assert_close(demb.cpu().double(), er.grad, rtol=rt, atol=tol["atol"] * 10)


@pytest.mark.parametrize("B,C,H,film,silu,add", [(3, 192, 24, True, True, True), (2, 576, 16, False, True, False), (2, 1536, 8, True, False, True),
                                                 (5, 96, 20, True, True, False), (2, 384, 23, False, False, True), (1, 128, 33, True, True, True)])
def test_groupnorm_flat_mapping_kernels(B, C, H, film, silu, add):
    """The bf16 GroupNorm passes of large launches run on the flat 16-byte kernels (unet_ops.hip: gns_*; chosen when a launch has >=
    512 workgroups, so the small shapes of the other tests never reach them): forced here (vaw_debug_gn_flat 1 / 0) on shapes with
    ragged chunk tails (HW % 128 != 0), 216 / 240 / 192 live lanes, groups that straddle a lane's channel octet, against the f64 torch
    GroupNorm and against the quad-mapped kernels."""
    HW = H * H
    x = (_rand(B, C, H, H, seed=1) * 1.5 + 0.3).bfloat16()
    gamma, beta = _rand(C, seed=2) * 0.5 + 1, _rand(C, seed=3) * 0.2
    emb = _rand(B, 3 * C, seed=4) * 0.3
    dout, dadd = _rand(B, C, H, H, seed=5).bfloat16(), _rand(B, C, H, H, seed=6).bfloat16()
    xr = x.double().requires_grad_(True)
    gr, br, er = gamma.double().requires_grad_(True), beta.double().requires_grad_(True), emb.double().requires_grad_(True)
    ref = F.group_norm(xr, 32, gr, br, eps=1e-5)
    if film:
        ref = ref * (1 + er[:, C:2 * C, None, None]) + er[:, 2 * C:, None, None]
    if silu:
        ref = F.silu(ref)
    (ref * dout.double()).sum().backward()
    xd, gd, bd, ed = _nhwc(x).to(DEV), gamma.to(DEV), beta.to(DEV), emb.to(DEV)
    dod, dad = _nhwc(dout).to(DEV), _nhwc(dadd).to(DEV)
    sc = ptr(ed) + 4 * C if film else None
    sh = ptr(ed) + 8 * C if film else None
    ws = torch.empty(lib().vaw_groupnorm_workspace_floats(B, HW, C), device=DEV)
    got = {}
    try:
        lib().vaw_debug_gn_coop(0)
        for mode in (1, 0):
            lib().vaw_debug_gn_flat(mode)
            ws.fill_(float("nan"))
            y = torch.empty(B * HW, C, device=DEV, dtype=torch.bfloat16)
            mean, rstd = torch.empty(B * 32, device=DEV), torch.empty(B * 32, device=DEV)
            assert lib().vaw_groupnorm_fwd(BF16, ptr(xd), ptr(gd), ptr(bd), sc, sh, 3 * C, int(silu), ptr(y), ptr(mean), ptr(rstd), B, HW, C,
                                           32, 1e-5, ptr(ws), stream_ptr()) == 0
            dx = torch.empty_like(xd)
            dg, db, demb = torch.ones(C, device=DEV), torch.ones(C, device=DEV), torch.zeros(B, 3 * C, device=DEV)
            ws.fill_(float("nan"))
            assert lib().vaw_groupnorm_bwd(BF16, ptr(dod), ptr(xd), ptr(mean), ptr(rstd), ptr(gd), ptr(bd), sc, sh, 3 * C, int(silu),
                                           ptr(dad) if add else None, ptr(dx), ptr(dg), ptr(db), 1.0, (ptr(demb) + 4 * C) if film else None,
                                           (ptr(demb) + 8 * C) if film else None, 3 * C, B, HW, C, 32, ptr(ws), stream_ptr()) == 0
            torch.cuda.synchronize()
            got[mode] = [t.cpu().double() for t in (y, mean, rstd, dx, dg, db, demb)]
    finally:
        lib().vaw_debug_gn_flat(-1)
        lib().vaw_debug_gn_coop(-1)
    tol = dict(rtol=3e-2, atol=3e-2)
    want_dx = xr.grad + (dadd.double() if add else 0)
    for mode in (1, 0):
        y, mean, rstd, dx, dg, db, demb = got[mode]
        torch.testing.assert_close(_nchw(y, B, H, H), ref.detach(), **tol)
        torch.testing.assert_close(_nchw(dx, B, H, H), want_dx, **tol)
        torch.testing.assert_close(dg, 1 + gr.grad, rtol=3e-2, atol=0.3)
        torch.testing.assert_close(db, 1 + br.grad, rtol=3e-2, atol=0.3)
        if film:
            torch.testing.assert_close(demb, er.grad, rtol=3e-2, atol=0.3)
    torch.testing.assert_close(got[1][1], got[0][1], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(got[1][2], got[0][2], rtol=1e-5, atol=1e-5)
    for k in (4, 5, 6):
        torch.testing.assert_close(got[1][k], got[0][k], rtol=2e-4, atol=2e-3)
    for k in (0, 3):
        assert float(((got[1][k] - got[0][k]).abs() > 0).double().mean()) < 0.05
        torch.testing.assert_close(got[1][k], got[0][k], rtol=1.6e-2, atol=1e-3)


@pytest.mark.parametrize("B,C,H,film,silu,add", [(3, 192, 32, True, True, True), (2, 576, 16, False, True, False), (2, 1536, 8, True, False, True),
                                                 (5, 96, 20, True, True, False), (2, 384, 24, False, False, True), (1, 128, 33, True, True, True),
                                                 (40, 64, 12, True, True, True)])
def test_groupnorm_cooperative_single_read_kernels(B, C, H, film, silu, add):
    """bf16 GroupNorm32 runs on the cooperative kernels (csrc/groupnorm_coop.h: every input read once, a sample's chunks held in
    the registers of several workgroups that exchange partial sums).  Same inputs through them and through the streaming kernels
    (vaw_debug_gn_coop 1 / 0): both within bf16 rounding of the f64 torch GroupNorm, statistics and parameter gradients within f32
    summation-order noise of each other.  Shapes: several chunks per sample with a ragged last one, groups that straddle a lane's
    channel octet (C/32 = 3, 6, 12, 18), workgroups of 480 / 504 live lanes, one chunk per sample (no wait), more samples than the
    grid holds at once."""
    HW = H * H
    x = (_rand(B, C, H, H, seed=1) * 1.5 + 0.3).bfloat16()
    gamma, beta = _rand(C, seed=2) * 0.5 + 1, _rand(C, seed=3) * 0.2
    emb = _rand(B, 3 * C, seed=4) * 0.3
    dout, dadd = _rand(B, C, H, H, seed=5).bfloat16(), _rand(B, C, H, H, seed=6).bfloat16()
    xr = x.double().requires_grad_(True)
    gr, br, er = gamma.double().requires_grad_(True), beta.double().requires_grad_(True), emb.double().requires_grad_(True)
    ref = F.group_norm(xr, 32, gr, br, eps=1e-5)
    if film:
        ref = ref * (1 + er[:, C:2 * C, None, None]) + er[:, 2 * C:, None, None]
    if silu:
        ref = F.silu(ref)
    (ref * dout.double()).sum().backward()
    xd, gd, bd, ed = _nhwc(x).to(DEV), gamma.to(DEV), beta.to(DEV), emb.to(DEV)
    dod, dad = _nhwc(dout).to(DEV), _nhwc(dadd).to(DEV)
    sc = ptr(ed) + 4 * C if film else None
    sh = ptr(ed) + 8 * C if film else None
    ws = torch.empty(lib().vaw_groupnorm_workspace_floats(B, HW, C), device=DEV)
    got = {}
    try:
        for mode in (1, 0):
            lib().vaw_debug_gn_coop(mode)
            ws.fill_(float("nan"))
            y = torch.empty(B * HW, C, device=DEV, dtype=torch.bfloat16)
            mean, rstd = torch.empty(B * 32, device=DEV), torch.empty(B * 32, device=DEV)
            assert lib().vaw_groupnorm_fwd(BF16, ptr(xd), ptr(gd), ptr(bd), sc, sh, 3 * C, int(silu), ptr(y), ptr(mean), ptr(rstd), B, HW, C,
                                           32, 1e-5, ptr(ws), stream_ptr()) == 0
            dx = torch.empty_like(xd)
            dg, db, demb = torch.ones(C, device=DEV), torch.ones(C, device=DEV), torch.zeros(B, 3 * C, device=DEV)
            ws.fill_(float("nan"))
            assert lib().vaw_groupnorm_bwd(BF16, ptr(dod), ptr(xd), ptr(mean), ptr(rstd), ptr(gd), ptr(bd), sc, sh, 3 * C, int(silu),
                                           ptr(dad) if add else None, ptr(dx), ptr(dg), ptr(db), 1.0, (ptr(demb) + 4 * C) if film else None,
                                           (ptr(demb) + 8 * C) if film else None, 3 * C, B, HW, C, 32, ptr(ws), stream_ptr()) == 0
            torch.cuda.synchronize()
            got[mode] = [t.cpu().double() for t in (y, mean, rstd, dx, dg, db, demb)]
    finally:
        lib().vaw_debug_gn_coop(-1)
    tol = dict(rtol=3e-2, atol=3e-2)
    want_dx = xr.grad + (dadd.double() if add else 0)
    for mode in (1, 0):
        y, mean, rstd, dx, dg, db, demb = got[mode]
        torch.testing.assert_close(_nchw(y, B, H, H), ref.detach(), **tol)
        torch.testing.assert_close(_nchw(dx, B, H, H), want_dx, **tol)
        torch.testing.assert_close(dg, 1 + gr.grad, rtol=3e-2, atol=0.3)
        torch.testing.assert_close(db, 1 + br.grad, rtol=3e-2, atol=0.3)
        if film:
            torch.testing.assert_close(demb, er.grad, rtol=3e-2, atol=0.3)
    # the two paths against each other: statistics to f32 noise, parameter gradients to summation-order noise of bf16-input sums
    torch.testing.assert_close(got[1][1], got[0][1], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(got[1][2], got[0][2], rtol=1e-5, atol=1e-5)
    for k in (4, 5, 6):
        torch.testing.assert_close(got[1][k], got[0][k], rtol=2e-4, atol=2e-3)
    # outputs: the same bf16 value except where the f32 results fall on two sides of a rounding boundary
    for k in (0, 3):
        diff = (got[1][k] - got[0][k]).abs()
        assert float((diff > 0).double().mean()) < 0.05
        torch.testing.assert_close(got[1][k], got[0][k], rtol=1.6e-2, atol=1e-3)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,C,H,W", [(2, 8, 5, 7), (1, 3, 6, 6), (2, 64, 8, 8)])
def test_im2col_col2im_match_unfold_fold(dtype, B, C, H, W):
    x = _rand(B, C, H, W, seed=1).to(dtype)
    dt = F32 if dtype == torch.float32 else BF16
    xd = _nhwc(x).to(DEV)
    col = torch.empty(B * H * W, 9 * C, device=DEV, dtype=dtype)
    assert lib().vaw_im2col3x3(dt, ptr(xd), ptr(col), B, H, W, C, stream_ptr()) == 0
    # F.unfold orders columns (c, kh, kw); ours (kh, kw, c)
    ref = F.unfold(x.float(), 3, padding=1).view(B, C, 9, H * W).permute(0, 3, 2, 1).reshape(B * H * W, 9 * C)
    assert torch.equal(col.cpu().float(), ref)
    dcol = _rand(B * H * W, 9 * C, seed=2).to(dtype)
    dx = torch.empty(B * H * W, C, device=DEV, dtype=dtype)
    dcd = dcol.to(DEV)
    assert lib().vaw_col2im3x3(dt, ptr(dcd), ptr(dx), B, H, W, C, stream_ptr()) == 0
    folded = F.fold(dcol.double().view(B, H * W, 9, C).permute(0, 3, 2, 1).reshape(B, C * 9, H * W), (H, W), 3, padding=1)
    tol = dict(rtol=1e-6, atol=1e-6) if dtype == torch.float32 else dict(rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(_nchw(dx.cpu().double(), B, H, W), folded, **tol)


def test_conv3x3_via_gemm_matches_torch_and_weight_layout():
    """conv = im2col + GEMM against the weight stored channels-last ([Co][3][3][Ci])."""
    B, Ci, Co, H = 2, 16, 24, 6
    x, w, b = _rand(B, Ci, H, H, seed=1), _rand(Co, Ci, 3, 3, seed=2) * 0.2, _rand(Co, seed=3)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    xd = _nhwc(x).to(DEV)
    col = torch.empty(B * H * H, 9 * Ci, device=DEV)
    lib().vaw_im2col3x3(F32, ptr(xd), ptr(col), B, H, H, Ci, stream_ptr())
    wk = w.permute(0, 2, 3, 1).reshape(Co, 9 * Ci).contiguous().to(DEV)
    y = ops.gemm_t(col, wk, bias=b.to(DEV))
    torch.testing.assert_close(_nchw(y.cpu().double(), B, H, H), ref, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("tile", [-1, 2, 3])      # kernel by shape / forced persistent kernel with 256- / 192-column tiles
@pytest.mark.parametrize("B,H,W,Ci,Co", [(2, 8, 8, 64, 128), (3, 8, 8, 128, 192), (1, 16, 8, 192, 64), (4, 4, 4, 64, 64),
                                         (8, 16, 16, 64, 192), (2, 32, 32, 128, 192), (3, 8, 24, 192, 384)])
def test_implicit_gemm_conv3x3_fwd_dgrad_wgrad(B, H, W, Ci, Co, tile):
    """vaw_conv3x3 (no patch matrix; padding taps read a zero page) vs torch.conv2d and its gradients.  Small-integer
    data makes the f32 weight gradient exact, so a wrong tap/pixel/channel address shows as a wrong integer."""
    lib().vaw_debug_gemm_tile(tile)
    try:
        _implicit_conv_case(B, H, W, Ci, Co)
    finally:
        lib().vaw_debug_gemm_tile(-1)


def _implicit_conv_case(B, H, W, Ci, Co):
    g = torch.Generator().manual_seed(B * H + Ci)
    x = torch.randint(-2, 3, (B, Ci, H, W), generator=g).float()
    w = torch.randint(-1, 2, (Co, Ci, 3, 3), generator=g).float()
    dy = torch.randint(-2, 3, (B, Co, H, W), generator=g).float()
    bias = torch.randint(-3, 4, (Co,), generator=g).float()
    res = torch.randint(-2, 3, (B, Co, H, W), generator=g).float()
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    ref = F.conv2d(xr, wr, bias.double(), padding=1) + res.double()
    (ref * dy.double()).sum().backward()
    bf = torch.bfloat16
    xd, dyd, rd = _nhwc(x).to(bf).to(DEV), _nhwc(dy).to(bf).to(DEV), _nhwc(res).to(bf).to(DEV)
    wd = w.permute(0, 2, 3, 1).contiguous().to(bf).to(DEV)            # stored [Co][3][3][Ci]
    bd = bias.to(DEV)
    M = B * H * W
    y = torch.empty(M, Co, device=DEV, dtype=bf)
    assert ops.conv3x3(BF16, 0, ptr(xd), None, ptr(wd), ptr(y), B, H, W, Ci, Co, bias=ptr(bd), resid=ptr(rd))
    torch.testing.assert_close(_nchw(y.cpu().double(), B, H, W), ref.detach(), rtol=1e-2, atol=1.0)   # bf16 output rounding only
    dx = torch.empty(M, Ci, device=DEV, dtype=bf)
    assert ops.conv3x3(BF16, 1, ptr(dyd), None, ptr(wd), ptr(dx), B, H, W, Ci, Co)
    torch.testing.assert_close(_nchw(dx.cpu().double(), B, H, W), xr.grad, rtol=1e-2, atol=1.0)
    dw = torch.ones(Co, 3, 3, Ci, device=DEV)
    assert ops.conv3x3(BF16, 2, ptr(dyd), ptr(xd), None, ptr(dw), B, H, W, Ci, Co, beta=1.0)
    assert torch.equal(dw.cpu().double().permute(0, 3, 1, 2), 1.0 + wr.grad)                           # exact, and beta=1 accumulated
    # bias gradient fused into the weight-gradient launch (row sums of the staged dy tiles), beta accumulates
    dw2, db = torch.zeros(Co, 3, 3, Ci, device=DEV), torch.full((Co,), 2.0, device=DEV)
    assert ops.conv3x3(BF16, 2, ptr(dyd), ptr(xd), None, ptr(dw2), B, H, W, Ci, Co, rowsum_a_out=ptr(db), rowsum_a_beta=0.5)
    assert torch.equal(dw2.cpu().double().permute(0, 3, 1, 2), wr.grad)
    assert torch.equal(db.cpu().double(), 1.0 + dy.double().sum((0, 2, 3)))
    # unsupported shapes decline without launching -- visibly: a one-time warning and a counter
    ops.fallbacks.clear()
    with pytest.warns(RuntimeWarning, match="falling back to im2col"):
        assert not ops.conv3x3(BF16, 0, ptr(xd), None, ptr(wd), ptr(y), B, H, W, 24, Co)
    assert not ops.conv3x3(BF16, 0, ptr(xd), None, ptr(wd), ptr(y), B, H, W, 24, Co)
    assert ops.fallbacks == {("conv3x3", "fwd", B, H, W, 24, Co): 2}
    assert not ops.conv3x3(F32, 0, ptr(xd), None, ptr(wd), ptr(y), B, H, W, Ci, Co)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,H,W,Cn,Cw", [(2, 8, 8, 3, 64), (1, 5, 7, 4, 320), (3, 4, 4, 1, 8), (2, 16, 16, 3, 192)])
def test_conv3x3_narrow_side(dtype, B, H, W, Cn, Cw):
    """vaw_conv3x3_narrow: the 3-channel stem (narrow in), the 3-channel output conv (narrow out) and that conv's
    input gradient, vs torch.conv2d.  Small integers: exact in bf16 operands, f32 accumulation."""
    dt = F32 if dtype == torch.float32 else BF16
    g = torch.Generator().manual_seed(B * H + Cw + Cn)
    ri = lambda *s, lo=-2, hi=3: torch.randint(lo, hi, s, generator=g).float()
    M = B * H * W
    # narrow in -> wide out (stem)
    x, w, bias = ri(B, Cn, H, W), ri(Cw, Cn, 3, 3, lo=-1, hi=2), ri(Cw)
    ref = F.conv2d(x.double(), w.double(), bias.double(), padding=1)
    xd, wd, bd = _nhwc(x).to(dtype).to(DEV), w.permute(0, 2, 3, 1).contiguous().to(dtype).to(DEV), bias.to(DEV)
    y = torch.empty(M, Cw, device=DEV, dtype=dtype)
    assert lib().vaw_conv3x3_narrow(dt, 0, ptr(xd), ptr(wd), ptr(bd), ptr(y), B, H, W, Cn, Cw, stream_ptr()) == 0
    assert torch.equal(_nchw(y.cpu().double(), B, H, W), ref)
    # wide in -> narrow out (output conv) and its input gradient
    x2, w2, b2, dy2 = ri(B, Cw, H, W, lo=-1, hi=2), ri(Cn, Cw, 3, 3, lo=-1, hi=2), ri(Cn), ri(B, Cn, H, W)
    xr = x2.double().requires_grad_(True)
    ref2 = F.conv2d(xr, w2.double(), b2.double(), padding=1)
    (ref2 * dy2.double()).sum().backward()
    x2d, w2d, b2d = _nhwc(x2).to(dtype).to(DEV), w2.permute(0, 2, 3, 1).contiguous().to(dtype).to(DEV), b2.to(DEV)
    y2 = torch.empty(M, Cn, device=DEV, dtype=dtype)
    assert lib().vaw_conv3x3_narrow(dt, 2, ptr(x2d), ptr(w2d), ptr(b2d), ptr(y2), B, H, W, Cn, Cw, stream_ptr()) == 0
    tol = dict(rtol=0, atol=0) if dtype == torch.float32 else dict(rtol=1e-2, atol=0.5)    # bf16 output rounding of sums > 256
    torch.testing.assert_close(_nchw(y2.cpu().double(), B, H, W), ref2.detach(), **tol)
    dy2d = _nhwc(dy2).to(dtype).to(DEV)
    dx2 = torch.empty(M, Cw, device=DEV, dtype=dtype)
    assert lib().vaw_conv3x3_narrow(dt, 1, ptr(dy2d), ptr(w2d), None, ptr(dx2), B, H, W, Cn, Cw, stream_ptr()) == 0
    assert torch.equal(_nchw(dx2.cpu().double(), B, H, W), xr.grad)
    # declines (nothing launched) off its shapes
    assert lib().vaw_conv3x3_narrow(dt, 0, ptr(xd), ptr(wd), ptr(bd), ptr(y), B, H, W, 5, Cw, stream_ptr()) == -3
    assert lib().vaw_conv3x3_narrow(dt, 0, ptr(xd), ptr(wd), ptr(bd), ptr(y), B, H, W, Cn, Cw + 4, stream_ptr()) == -3


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_resample_concat_layout(dtype):
    B, C, H = 2, 8, 6
    dt = F32 if dtype == torch.float32 else BF16
    x = _rand(B, C, H, H, seed=1).to(dtype)
    xd = _nhwc(x).to(DEV)
    dn = torch.empty(B * (H // 2) ** 2, C, device=DEV, dtype=dtype)
    assert lib().vaw_resample2(dt, ptr(xd), ptr(dn), B, H // 2, H // 2, C, 0, 0.25, stream_ptr()) == 0
    torch.testing.assert_close(_nchw(dn.cpu().float(), B, H // 2, H // 2), F.avg_pool2d(x.float(), 2), rtol=1e-2, atol=1e-2)
    up = torch.empty(B * (2 * H) ** 2, C, device=DEV, dtype=dtype)
    assert lib().vaw_resample2(dt, ptr(xd), ptr(up), B, 2 * H, 2 * H, C, 1, 1.0, stream_ptr()) == 0
    assert torch.equal(_nchw(up.cpu().float(), B, 2 * H, 2 * H), F.interpolate(x.float(), scale_factor=2, mode="nearest"))
    a, b = _rand(10, 8, seed=2).to(dtype).to(DEV), _rand(10, 12, seed=3).to(dtype).to(DEV)
    cat = torch.empty(10, 20, device=DEV, dtype=dtype)
    assert lib().vaw_concat_channels(dt, ptr(a), ptr(b), ptr(cat), 10, 8, 12, 0, stream_ptr()) == 0
    assert torch.equal(cat, torch.cat([a, b], 1))
    a2, b2 = torch.empty_like(a), torch.empty_like(b)
    assert lib().vaw_concat_channels(dt, ptr(a2), ptr(b2), ptr(cat), 10, 8, 12, 1, stream_ptr()) == 0
    assert torch.equal(a2, a) and torch.equal(b2, b)
    xin = _rand(B, 3, H, H, seed=4).to(DEV)
    nh = torch.empty(B * H * H, 3, device=DEV, dtype=dtype)
    assert lib().vaw_nchw_to_nhwc(dt, ptr(xin), ptr(nh), B, 3, H * H, stream_ptr()) == 0
    back = torch.empty_like(xin)
    assert lib().vaw_nhwc_to_nchw(dt, ptr(nh), ptr(back), B, 3, H * H, stream_ptr()) == 0
    torch.testing.assert_close(back, xin, rtol=1e-2 if dtype == torch.bfloat16 else 0, atol=1e-2 if dtype == torch.bfloat16 else 0)


def test_data_movement_fast_paths_bf16_exact():
    """bf16 with channel counts on the 8 grid takes the 16-byte kernels (pool / nearest x2, concat / split, gradient add): ragged
    totals (the last workgroup's octets run out mid-tile), every result bit for bit what the 8-byte kernels' arithmetic gives."""
    B, C, H, W = 3, 40, 10, 14
    x = _rand(B, C, H, W, seed=1).bfloat16()
    xd = _nhwc(x).to(DEV)
    dn = torch.empty(B * (H // 2) * (W // 2), C, device=DEV, dtype=torch.bfloat16)
    assert lib().vaw_resample2(BF16, ptr(xd), ptr(dn), B, H // 2, W // 2, C, 0, 0.25, stream_ptr()) == 0
    xf = x.float()
    want = ((((xf[:, :, 0::2, 0::2] + xf[:, :, 0::2, 1::2]) + xf[:, :, 1::2, 0::2]) + xf[:, :, 1::2, 1::2]) * 0.25).bfloat16()
    assert torch.equal(_nchw(dn.cpu(), B, H // 2, W // 2), want)
    up = torch.empty(B * 4 * H * W, C, device=DEV, dtype=torch.bfloat16)
    assert lib().vaw_resample2(BF16, ptr(xd), ptr(up), B, 2 * H, 2 * W, C, 1, 1.0, stream_ptr()) == 0
    assert torch.equal(_nchw(up.cpu().float(), B, 2 * H, 2 * W), F.interpolate(xf, scale_factor=2, mode="nearest"))
    M = 1237
    a, b = _rand(M, 24, seed=2).bfloat16().to(DEV), _rand(M, 40, seed=3).bfloat16().to(DEV)
    cat = torch.empty(M, 64, device=DEV, dtype=torch.bfloat16)
    assert lib().vaw_concat_channels(BF16, ptr(a), ptr(b), ptr(cat), M, 24, 40, 0, stream_ptr()) == 0
    assert torch.equal(cat, torch.cat([a, b], 1))
    a2, b2 = torch.zeros_like(a), torch.zeros_like(b)
    assert lib().vaw_concat_channels(BF16, ptr(a2), ptr(b2), ptr(cat), M, 24, 40, 1, stream_ptr()) == 0
    assert torch.equal(a2, a) and torch.equal(b2, b)
    n = 8 * 12345
    d0, sr = _rand(n, seed=4).bfloat16().to(DEV), _rand(n, seed=5).bfloat16().to(DEV)
    d = torch.cat([d0, torch.full((64,), 7.0, device=DEV, dtype=torch.bfloat16)])          # canary behind the buffer
    assert lib().vaw_add_inplace(BF16, ptr(d), ptr(sr), n, stream_ptr()) == 0
    assert torch.equal(d[:n], (d0.float() + sr.float()).bfloat16()) and bool((d[n:] == 7.0).all())


@pytest.mark.parametrize("tag", ["new", "legacy_ss", "legacy"])     # legacy: use_scale_shift_norm=False, resblock_updown=False (conv
def test_unet_tiny_fp32_matches_reference_golden(tag):                 # Upsample / stride-2 Downsample), legacy attention order
    g = load_pt("unet_tiny.pt")
    torch.manual_seed(21)
    m = vaw_amd.UNetModel(compute_dtype="fp32", **g[f"{tag}/kw"])
    assert_fingerprints(m.state_dict(), g[f"{tag}/init_sd"], 1e-6, 1e-9, "same seed => the reference's initial weights")
    m = m.to(DEV).train()
    perturb_(m, 77, std=0.03)
    x, t, gout = (g[f"{tag}/{k}"].to(DEV) for k in ("x", "t", "gout"))
    y = g[f"{tag}/y"].to(DEV) if g[f"{tag}/y"].numel() else None
    xr = x.clone().requires_grad_(True)
    out = m(xr, t, y=y) if y is not None else m(xr, t)
    (out * gout).sum().backward()
    torch.testing.assert_close(out.detach().cpu(), g[f"{tag}/out"], rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(xr.grad.cpu(), g[f"{tag}/gx"], rtol=1e-4, atol=2e-5)
    grads = {k: p.grad.cpu() for k, p in m.named_parameters() if p.grad is not None}
    assert_fingerprints(grads, g[f"{tag}/grads"], 1e-4, 2e-5, "parameter gradients")
    # second backward accumulates (torch convention)
    g1 = {k: v.clone() for k, v in grads.items()}
    out = m(xr, t, y=y) if y is not None else m(xr, t)
    (out * gout).sum().backward()
    for k, p in m.named_parameters():
        if p.grad is not None:
            torch.testing.assert_close(p.grad.cpu(), 2 * g1[k], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_unet_tiny_dropout_vs_reference_golden(dtype):
    """--dropout (reference main.py:99, models/unet.py:206-213): nn.Dropout(0.1) inside every ResBlock.  The keep masks come from
    the CPU generator in the reference's order (host_dropout_rng), so the training-mode forward / backward of the unmodified
    reference under the same seed is reproduced; eval mode must ignore dropout."""
    g = load_pt("unet_dropout.pt")
    torch.manual_seed(21)
    m = vaw_amd.UNetModel(compute_dtype=dtype, **g["kw"])
    m = m.to(DEV).train()
    perturb_(m, 77, std=0.03)
    m.host_dropout_rng = True
    x, t, gout, y = (g[k].to(DEV) for k in ("x", "t", "gout", "y"))
    xr = x.clone().requires_grad_(True)
    torch.manual_seed(5)
    out = m(xr, t, y=y)
    (out * gout).sum().backward()
    grads = {k: p.grad.cpu() for k, p in m.named_parameters() if p.grad is not None}
    if dtype == "fp32":
        torch.testing.assert_close(out.detach().cpu(), g["out"], rtol=1e-4, atol=2e-5)
        torch.testing.assert_close(xr.grad.cpu(), g["gx"], rtol=1e-4, atol=2e-5)
        assert_fingerprints(grads, g["grads"], 1e-4, 2e-5, "parameter gradients")
    else:
        assert float((out.detach().cpu() - g["out"]).norm() / g["out"].norm()) < 4e-2
    m.eval()
    with torch.no_grad():
        oe = m(x, t, y=y)
    if dtype == "fp32":
        torch.testing.assert_close(oe.cpu(), g["out_eval"], rtol=1e-4, atol=2e-5)
    # device-RNG masks (throughput mode): about 10 % of the activations dropped, different draws give different outputs
    m.train()
    m.host_dropout_rng = False
    o1, o2 = m(x, t, y=y).detach(), m(x, t, y=y).detach()
    assert not torch.equal(o1, o2) and torch.isfinite(o1).all()


def test_unet_tiny_bf16_close_to_reference():
    g = load_pt("unet_tiny.pt")
    tag = "new"
    torch.manual_seed(21)
    m = vaw_amd.UNetModel(compute_dtype="bf16", **g[f"{tag}/kw"]).to(DEV).train()
    perturb_(m, 77, std=0.03)
    x, t, gout, y = (g[f"{tag}/{k}"].to(DEV) for k in ("x", "t", "gout", "y"))
    out = m(x, t, y=y)
    (out * gout).sum().backward()
    ref = g[f"{tag}/out"]
    assert float((out.detach().cpu() - ref).norm() / ref.norm()) < 4e-2
    for k, p in m.named_parameters():
        if p.grad is not None:
            gl2 = float(g[f"{tag}/grads"][k]["stats"][2])
            if gl2 < 1e-2:
                continue      # e.g. a conv bias feeding GroupNorm: its exact gradient is 0, bf16 leaves rounding noise
            assert abs(float(p.grad.double().norm()) - gl2) <= 8e-2 * gl2 + 1e-3, k


def _run_trainer(model, args, batches, steps, fused, var_type="FIXED_LARGE"):
    ema_model = copy.deepcopy(model)
    if fused:
        opt = vaw_amd.FusedAdamW(model, lr=args.lr, betas=(0.9, 0.95), weight_decay=0.0, eps=1e-8)
    else:
        opt = torch.optim.AdamW(model.parameters(), lr=args.lr, betas=(0.9, 0.95), weight_decay=0.0, eps=1e-8)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=vaw_amd.get_lr_lambda(args))
    diff = vaw_amd.GaussianDiffusion(args=args, betas=vaw_amd.get_named_beta_schedule(args.path_type, 1000),
                                     model_mean_type=vaw_amd.ModelMeanType.EPSILON,
                                     model_var_type=vaw_amd.ModelVarType[var_type], loss_type=vaw_amd.LossType.MSE,
                                     rescale_timesteps=True)
    tr = vaw_amd.Trainer(args, torch.device(DEV), model, ema_model, opt, sched, diff, batches, Pbar())
    losses = [tr.train_step(s) for s in range(1, steps + 1)]
    psum = float(sum(p.double().abs().sum() for p in model.parameters()))
    esum = float(sum(v.double().abs().sum() for v in ema_model.state_dict().values()))
    return losses, psum, esum


def test_unet_bf16_hip_graph_step_follows_eager_step():
    """A bf16 UNet with 1 x 1 layers (attention qkv / proj_out, skip connections on channel changes): the eager step defers their
    weight gradients into grouped launches (unet.py: _flush_wgrads), a captured step cannot (a new group uploads a descriptor table)
    and takes the per-layer launches.  Both must train, on the same trajectory up to the summation order of those weight gradients;
    with VAW_UNET_GROUPED_WGRAD=0 semantics (no deferral at all) eager and captured steps are the same kernels.  (A capture that
    reached the upload used to abort the step: this is its regression test.)"""
    class FixedDraws(vaw_amd.GaussianDiffusion):
        def training_losses(self, model, x_start, features=None, t=None, model_kwargs=None, noise=None):
            return super().training_losses(model, x_start, features, t=self._t, model_kwargs=model_kwargs, noise=self._noise)

    def run(graph, grouped):
        args = base_args(image_size=16, lr=1e-3, grad_clip=0.5, defer_loss_sync=True, hip_graph=graph)
        random.seed(42); np.random.seed(42); torch.manual_seed(42)
        model = vaw_amd.UNetModel(16, 3, 32, 3, 1, attention_resolutions=(1, 2), channel_mult=(1, 2), num_heads=2, use_scale_shift_norm=True,
                                  resblock_updown=True, use_new_attention_order=True, compute_dtype="bf16").to(DEV)
        perturb_(model, 5)
        model._grouped_wgrad = grouped
        ema_model = copy.deepcopy(model)
        opt = vaw_amd.FusedAdamW(model, lr=args.lr, betas=(0.9, 0.95), weight_decay=0.0, eps=1e-8)
        sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=vaw_amd.get_lr_lambda(args))
        diff = FixedDraws(args=args, betas=vaw_amd.get_named_beta_schedule("cosine", 1000), model_mean_type=vaw_amd.ModelMeanType.EPSILON,
                          model_var_type=vaw_amd.ModelVarType.FIXED_LARGE, loss_type=vaw_amd.LossType.MSE, rescale_timesteps=True)
        g = torch.Generator().manual_seed(9)
        diff._t = torch.randint(0, 1000, (8,), generator=g).to(DEV)
        diff._noise = torch.randn(8, 3, 16, 16, generator=g).to(DEV)
        batches = [(torch.randn(8, 3, 16, 16, generator=g), torch.zeros(8, dtype=torch.long)) for _ in range(3)]
        tr = vaw_amd.Trainer(args, torch.device(DEV), model, ema_model, opt, sched, diff, batches, Pbar())
        losses = [float(tr.train_step(s)) for s in range(1, 9)]
        return losses, model._flat.clone()

    le, pe = run(False, True)
    lg, pg = run(True, True)
    assert all(np.isfinite(le)) and all(np.isfinite(lg)) and le[-1] < le[0] and lg[-1] < lg[0]
    np.testing.assert_allclose(lg, le, rtol=2e-2)
    assert float((pg - pe).norm() / pe.norm()) < 1e-3
    le0, pe0 = run(False, False)
    lg0, pg0 = run(True, False)
    assert le0 == lg0 and torch.equal(pe0, pg0)


CFG1 = lambda: vaw_amd.UNetModel(32, 3, 64, 3, 2, attention_resolutions=(), channel_mult=(1, 2, 2, 2), num_heads=4,
                                 use_scale_shift_norm=True, resblock_updown=True, use_new_attention_order=True,
                                 compute_dtype="fp32")


@pytest.mark.parametrize("name,args,steps,fused", [
    ("cfg1", base_args(cpu_rng=True), 5, True),
    ("cfg1", base_args(cpu_rng=True), 5, False),
    ("cfg1_accum2_clip", base_args(grad_accumulation=2, grad_clip=1.0, cpu_rng=True), 3, True),
    ("cfg1_warmup_cosine_minsnr", base_args(weight_type="min_snr_5", warmup_steps=2, cosine_decay=True, total_steps=10,
                                            final_lr=1e-5, cpu_rng=True), 4, True),
])
def test_trainer_trajectory_cfg1_unet_fp32_vs_reference(name, args, steps, fused):
    """BASELINE config 1 (CIFAR-10-shaped UNet, 10.4 M parameters, batch 16): the reference's CPU Trainer
    trajectories (tests/golden/trainer.json; cfg1 is the trajectory quoted in BASELINE.md §2) vs the HIP path in
    f32 parity mode with the CPU RNG stream injected.  Tolerance: north_star's 1e-4 relative per step."""
    exp = load_json("trainer.json")[name]
    random.seed(42); np.random.seed(42); torch.manual_seed(42)
    model = CFG1().to(DEV)
    losses, psum, esum = _run_trainer(model, args, synth_loader(16, 3, 32, 4, 0), steps, fused)
    np.testing.assert_allclose(losses, exp["losses"], rtol=1e-4)
    assert psum == pytest.approx(exp["param_abs_sum"], rel=1e-5)
    assert esum == pytest.approx(exp["ema_abs_sum"], rel=1e-6)


def test_trainer_trajectory_tiny_unet_learned_variance_vs_reference():
    """UNet with a 6-channel output (learn_sigma) + LEARNED_RANGE: loss = mse + vb; 5 reference steps, 1e-4 relative.
    Also covers attention at 8x8 inside the Trainer and an output conv that is neither narrow (<= 4) nor MFMA-shaped."""
    exp = load_json("trainer_vb.json")["unet_tiny_learn_sigma"]
    args = base_args(image_size=16, lr=1e-3, learn_sigma=True, cpu_rng=True)
    random.seed(42); np.random.seed(42); torch.manual_seed(42)
    model = vaw_amd.UNetModel(16, 3, 32, 6, 1, attention_resolutions=(2,), channel_mult=(1, 2), num_heads=2,
                              use_scale_shift_norm=True, resblock_updown=True, use_new_attention_order=True,
                              compute_dtype="fp32").to(DEV)
    losses, psum, esum = _run_trainer(model, args, synth_loader(8, 3, 16, 3, 0), 5, True, var_type="LEARNED_RANGE")
    np.testing.assert_allclose(losses, exp["losses"], rtol=1e-4)
    assert psum == pytest.approx(exp["param_abs_sum"], rel=1e-5)
    assert esum == pytest.approx(exp["ema_abs_sum"], rel=1e-6)


def test_unet_backward_stage_hooks_tile_the_gradient_buffer():
    """DDP buckets of the UNet: decoder (output_blocks + out) first, then middle_block, then the rest; the ranges tile the
    flat gradient buffer exactly once, and each stage's gradients are final (equal to the end-of-backward values) when
    its hook fires."""
    m = vaw_amd.UNetModel(16, 3, 32, 3, 1, attention_resolutions=(2,), channel_mult=(1, 2), num_heads=2, use_scale_shift_norm=True,
                          resblock_updown=True, use_new_attention_order=True, compute_dtype="fp32")
    perturb_(m, 3)
    m = m.to(DEV)
    m.ensure_flat()
    bounds = m.grad_stage_bounds()
    assert set(bounds) == {3, 2, 0}
    cover = torch.zeros(m._flat_n_train, dtype=torch.int32)
    for rng in bounds.values():
        for lo, hi in (rng if isinstance(rng, list) else [rng]):
            cover[lo:hi] += 1
    assert int(cover.min()) == 1 and int(cover.max()) == 1
    g = torch.Generator().manual_seed(2)
    x, t = torch.randn(2, 3, 16, 16, generator=g).to(DEV), (torch.rand(2, generator=g) * 999).to(DEV)
    gout = torch.randn(2, 3, 16, 16, generator=g).to(DEV)
    snaps = []
    m.grad_ready_hook = lambda st: snaps.append((st, m.flat_grads()[bounds[st][0]:bounds[st][1]].clone()))
    (m(x, t) * gout).sum().backward()
    m.grad_ready_hook = None
    assert [s for s, _ in snaps] == [3, 2, 0]
    final = m.flat_grads()
    for st, snap in snaps:
        assert torch.equal(snap, final[bounds[st][0]:bounds[st][1]]), st
        assert float(snap.abs().max()) > 0


@pytest.mark.parametrize("name,size,chans", [("LDM", 32, 4), ("ADM-32", 32, 3), ("UNet-32", 32, 3)])
def test_other_unet_factories_fp32_vs_oracle(name, size, chans):
    """The factories no BASELINE config trains (LDM: 4-channel latents, 32-channel heads, mult 1,2,4; ADM-32; UNet-32) on the
    HIP engine in f32 against the CPU oracle of the same architecture: output and gradients at batch 2."""
    from oracle import unet as ounet
    kw = dict(num_classes=10, class_cond=True)
    torch.manual_seed(3)
    ref = getattr(ounet, name.replace("-", "_"))(**kw)
    torch.manual_seed(3)
    m = getattr(vaw_amd.unet, name.replace("-", "_"))(compute_dtype="fp32", **kw)
    perturb_(ref, 77, std=0.02)
    perturb_(m, 77, std=0.02)
    m = m.to(DEV).train()
    ref.train()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, chans, size, size, generator=g)
    t = torch.tensor([17.0, 803.0])
    y = torch.tensor([3, 7])
    gout = torch.randn(2, chans, size, size, generator=g)
    out_r = ref(x, t, y=y)
    out_r = out_r[0] if isinstance(out_r, tuple) else out_r
    (out_r * gout).sum().backward()
    out = m(x.to(DEV), t.to(DEV), y=y.to(DEV))
    out = out[0] if isinstance(out, tuple) else out
    (out * gout.to(DEV)).sum().backward()
    torch.testing.assert_close(out.detach().cpu(), out_r.detach(), rtol=1e-4, atol=1e-5)
    gr = dict(ref.named_parameters())
    worst = 0.0
    for k, p in m.named_parameters():
        r = gr[k].grad
        if r is None or float(r.norm()) < 1e-8:
            continue
        worst = max(worst, float((p.grad.detach().cpu() - r).norm() / r.norm()))
    assert worst < 1e-3, worst
