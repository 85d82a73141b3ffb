"""BASELINE.json configs 2, 3 and 5 at full model size on the GPU: `UNet_64(class_cond=False)`, `ADM_64(num_classes=1000)` and
`DiT_XL(32, patch 2)` (reference models/unet.py:993,1013, models/dit.py:373).

  * f32 parity mode at batch 2 against tests/golden/bigcfg.pt (written by tests/golden/make_goldens.py from the unmodified
    reference): the seed-reconstructed weights, per-sample terms['mse'] (1e-4 relative, the north_star tolerance), every
    parameter gradient, and two `Trainer` steps;
  * bf16 throughput mode at batch 2: drift against the same fixture, asserted loosely and REPORTED;
  * the configs' full batches (128 / 256 / 128 per GPU) in bf16 through size-independent properties: determinism, per-sample
    independence, finite non-zero gradients.
Config 5's fp8 variant (compute_dtype="fp8": the blocks' Linear layers on the scaled fp8 MFMA) has no reference implementation to
pin against -- the reference trains DiT-XL/2 under bf16 autocast -- so it is held to the same f32 fixture with its own, looser,
REPORTED drift bounds, plus determinism and a short training run that must track the bf16 one."""
import copy
import random

import numpy as np
import pytest
import torch

from conftest import Pbar, assert_fingerprints, base_args, fingerprint, load_pt, perturb_, synth_loader

pytestmark = pytest.mark.gpu

import vaw_amd

DEV = "cuda"

CONFIGS = {
    "unet64": dict(kind="unet", make=lambda dt: vaw_amd.UNet_64(class_cond=False, compute_dtype=dt), size=64, chans=3, classes=0,
                   full_batch=128),
    "adm64": dict(kind="unet", make=lambda dt: vaw_amd.ADM_64(num_classes=1000, class_cond=True, compute_dtype=dt), size=64, chans=3,
                  classes=1000, full_batch=256),
    "dit_xl2": dict(kind="dit", make=lambda dt: vaw_amd.DiT_XL(image_size=32, patch_size=2, in_channels=4, class_dropout_prob=0.0,
                                                               num_classes=1000, learn_sigma=False, compute_dtype=dt),
                    size=32, chans=8, classes=1000, full_batch=128),
}


def _args(c, **kw):
    latent = c["kind"] == "dit"
    return base_args(in_chans=4 if latent else 3, class_cond=bool(c["classes"]), dataset="Latent" if latent else "ImageNet",
                     image_size=c["size"], **kw)


def _build(c, dtype):
    """Same construction as the fixture: seed 42, the reference constructor's RNG order, then perturb_(model, 7, 0.02)."""
    random.seed(42); np.random.seed(42); torch.manual_seed(42)
    m = c["make"](dtype)
    perturb_(m, 7, std=0.02)
    return m


def _diffusion(args):
    return vaw_amd.GaussianDiffusion(args=args, betas=vaw_amd.get_named_beta_schedule("cosine", 1000),
                                     model_mean_type=vaw_amd.ModelMeanType.EPSILON, model_var_type=vaw_amd.ModelVarType.FIXED_LARGE,
                                     loss_type=vaw_amd.LossType.MSE, rescale_timesteps=True)


def _objective(model, c, g):
    diff = _diffusion(_args(c))
    kw = {"y": g["y"].to(DEV)} if c["classes"] else {}
    terms = diff.training_losses(model, g["x"].to(DEV), None, t=g["t"].to(DEV), model_kwargs=kw, noise=g["noise"].to(DEV))
    terms["loss"].mean().backward()
    return terms["mse"].detach().double().cpu(), {k: p.grad.detach().cpu() for k, p in model.named_parameters() if p.grad is not None}


@pytest.mark.parametrize("name", list(CONFIGS))
def test_full_size_model_fp32_objective_and_gradients_vs_reference(name):
    c, g = CONFIGS[name], load_pt("bigcfg.pt")[name]
    m = _build(c, "fp32")
    assert sum(p.numel() for p in m.parameters()) == g["n_params"]
    assert_fingerprints({k: v.detach() for k, v in m.named_parameters()}, g["params"], 1e-6, 1e-9, "seed-reconstructed weights")
    m = m.to(DEV).train()
    mse, grads = _objective(m, c, g)
    torch.testing.assert_close(mse, g["mse"], rtol=1e-4, atol=0)
    # gradients: 1e-4 of each tensor's rms on the strided sample, 1e-3 on its l2 norm (f32 accumulation order differs)
    assert_fingerprints(grads, g["grads"], 1e-4, 2e-6, "parameter gradients")


@pytest.mark.parametrize("name", list(CONFIGS))
def test_full_size_model_fp32_trainer_steps_vs_reference(name):
    """Two reference `Trainer.train_step`s (AdamW 1e-4, EMA, CPU RNG stream injected) at batch 2: losses within 1e-4."""
    c, g = CONFIGS[name], load_pt("bigcfg.pt")[name]["trainer"]
    args = _args(c, cpu_rng=True)
    random.seed(42); np.random.seed(42); torch.manual_seed(42)
    model = c["make"]("fp32")
    perturb_(model, 7, std=0.02)
    model = model.to(DEV)
    ema_model = copy.deepcopy(model)
    opt = vaw_amd.FusedAdamW(model, lr=args.lr, betas=(0.9, 0.95), weight_decay=0.0, eps=1e-8)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=vaw_amd.get_lr_lambda(args))
    loader = synth_loader(2, c["chans"], c["size"], 2, c["classes"], latent=c["kind"] == "dit")
    tr = vaw_amd.Trainer(args, torch.device(DEV), model, ema_model, opt, sched, _diffusion(args), loader, Pbar())
    losses = [tr.train_step(s) for s in (1, 2)]
    np.testing.assert_allclose(losses, g["losses"], rtol=1e-4)
    psum = float(sum(p.double().abs().sum() for p in model.parameters()))
    esum = float(sum(v.double().abs().sum() for v in ema_model.state_dict().values()))
    assert psum == pytest.approx(g["param_abs_sum"], rel=1e-5)
    assert esum == pytest.approx(g["ema_abs_sum"], rel=1e-6)


@pytest.mark.parametrize("name", list(CONFIGS))
def test_full_size_model_bf16_drift_vs_reference(name, record_property):
    """bf16 MFMA kernels (the ones bench.py times) against the f32 reference fixture: per-sample loss within 2e-2 and every
    gradient tensor's l2 norm within 8e-2 of the reference's; the measured drift is reported (pytest -rA / junit properties)."""
    c, g = CONFIGS[name], load_pt("bigcfg.pt")[name]
    m = _build(c, "bf16").to(DEV).train()
    mse, grads = _objective(m, c, g)
    rel = ((mse - g["mse"]).abs() / g["mse"].abs()).max().item()
    worst, worst_k = 0.0, ""
    for k, v in grads.items():
        ref = float(g["grads"][k]["stats"][2])
        if ref > 1e-6:
            d = abs(float(v.double().norm()) - ref) / ref
            if d > worst:
                worst, worst_k = d, k
    record_property("bf16_mse_rel_drift", rel)
    record_property("bf16_worst_grad_l2_drift", f"{worst:.4f} ({worst_k})")
    print(f"[bf16 drift] {name}: per-sample mse {rel:.2e}, worst gradient l2 {worst:.2e} at {worst_k}")
    assert rel < 2e-2
    assert worst < 8e-2, worst_k


@pytest.mark.parametrize("name", list(CONFIGS))
def test_full_batch_properties_bf16(name):
    """The config's full per-GPU batch in bf16 (no oracle run at this size): determinism (bitwise equal losses and gradients
    on a repeat), per-sample independence (a sample's loss alone == its loss in the batch), finite non-zero gradients."""
    c = CONFIGS[name]
    B, Cx, S = c["full_batch"], (4 if c["kind"] == "dit" else 3), c["size"]
    m = _build(c, "bf16").to(DEV).train()
    diff = _diffusion(_args(c))
    g = torch.Generator().manual_seed(1)
    x0 = (torch.randn(B, Cx, S, S, generator=g) * 0.5).to(DEV)
    noise = torch.randn(B, Cx, S, S, generator=g).to(DEV)
    t = torch.randint(0, 1000, (B,), generator=g).to(DEV)
    y = torch.randint(0, 1000, (B,), generator=g).to(DEV)
    kw = (lambda idx: {"y": y[idx]}) if c["classes"] else (lambda idx: {})
    every = torch.arange(B, device=DEV)
    t1 = diff.training_losses(m, x0, None, t=t, model_kwargs=kw(every), noise=noise)
    t1["loss"].mean().backward()
    g1 = m.flat_grads().clone()
    m.zero_grad_flat()
    t2 = diff.training_losses(m, x0, None, t=t, model_kwargs=kw(every), noise=noise)
    t2["loss"].mean().backward()
    assert torch.equal(t1["mse"], t2["mse"]) and torch.equal(g1, m.flat_grads())
    idx = torch.tensor([1, B - 3], device=DEV)
    sub = diff.training_losses(m, x0[idx], None, t=t[idx], model_kwargs=kw(idx), noise=noise[idx])
    torch.testing.assert_close(sub["mse"], t1["mse"][idx], rtol=2e-3, atol=1e-5)
    assert torch.isfinite(g1).all() and float(g1.abs().max()) > 0 and torch.isfinite(t1["mse"]).all()


@pytest.mark.parametrize("gfmt", ["e5m2", "e4m3"])
def test_dit_xl2_fp8_drift_vs_reference(gfmt, record_property):
    """DiT-XL/2 with fp8 GEMMs (e4m3 weights / activations, `gfmt` gradients, per-tensor just-in-time scales) against the f32
    reference fixture at batch 2 (512 tokens: the least averaging a weight gradient ever gets): per-sample loss within 5e-2,
    every gradient tensor's l2 norm within 0.25, its direction (cosine on the fixture's 64-element strided sample) above 0.97 in
    the median and 0.5 at worst; measured values are reported."""
    c, g = CONFIGS["dit_xl2"], load_pt("bigcfg.pt")["dit_xl2"]
    m = _build(c, "fp8")
    m.fp8_grad_format, m.fp8_scaling = gfmt, "jit"        # scales measured on the tensors themselves in every pass
    m = m.to(DEV).train()
    mse, grads = _objective(m, c, g)
    rel = ((mse - g["mse"]).abs() / g["mse"].abs()).max().item()
    worst, worst_k, worst_cos, worst_cos_k, coss = 0.0, "", 1.0, "", []
    for k, v in grads.items():
        ref = g["grads"][k]
        l2 = float(ref["stats"][2])
        if l2 > 1e-6:
            d = abs(float(v.double().norm()) - l2) / l2
            if d > worst:
                worst, worst_k = d, k
            sample = fingerprint(v)[1]
            if sample.numel() >= 16 and float(ref["sample"].double().norm()) > 0:
                cos = float(torch.nn.functional.cosine_similarity(sample, ref["sample"].double(), dim=0))
                coss.append(cos)
                if cos < worst_cos:
                    worst_cos, worst_cos_k = cos, k
    record_property(f"fp8_{gfmt}_mse_rel_drift", rel)
    record_property(f"fp8_{gfmt}_worst_grad_l2_drift", f"{worst:.4f} ({worst_k})")
    record_property(f"fp8_{gfmt}_worst_grad_cosine", f"{worst_cos:.4f} ({worst_cos_k})")
    med = float(np.median(coss))
    record_property(f"fp8_{gfmt}_median_grad_cosine", med)
    print(f"[fp8/{gfmt} drift] dit_xl2: per-sample mse {rel:.2e}, worst gradient l2 {worst:.2e} at {worst_k}, gradient cosine on the "
          f"64-element samples: median {med:.4f}, worst {worst_cos:.4f} at {worst_cos_k}")
    assert rel < 5e-2
    assert worst < 0.25, worst_k
    assert med > 0.97 and worst_cos > 0.5, worst_cos_k
    # deterministic: a second pass reproduces losses and gradients bit for bit
    g1 = m.flat_grads().clone()
    m.zero_grad_flat()
    mse2, _ = _objective(m, c, g)
    assert torch.equal(mse, mse2) and torch.equal(g1, m.flat_grads())


def test_dit_fp8_training_tracks_bf16():
    """Thirty optimizer steps of a DiT-B/4-width model on a fixed batch, fp8 (default recipe: delayed scaling) vs bf16 from the
    same start: both must learn (loss falls by > 20 %), the fp8 loss curve must stay within 5 % of the bf16 one, and a repeat
    of the fp8 run must reproduce it exactly."""
    def run(dtype):
        random.seed(3); np.random.seed(3); torch.manual_seed(3)
        m = vaw_amd.DiT(image_size=32, patch_size=4, in_channels=4, hidden_size=768, depth=4, num_heads=12, class_dropout_prob=0.0,
                        num_classes=10, compute_dtype=dtype)
        perturb_(m, 7, std=0.02)
        m = m.to(DEV).train()
        opt = vaw_amd.FusedAdamW(m, lr=1e-3, betas=(0.9, 0.95), weight_decay=0.0, eps=1e-8)
        c = dict(kind="dit", classes=10, size=32)
        diff = _diffusion(_args(c))
        g = torch.Generator().manual_seed(1)
        B = 16
        x0 = (torch.randn(B, 4, 32, 32, generator=g) * 0.5).to(DEV)
        noise = torch.randn(B, 4, 32, 32, generator=g).to(DEV)
        t = torch.randint(0, 1000, (B,), generator=g).to(DEV)
        y = torch.randint(0, 10, (B,), generator=g).to(DEV)
        out = []
        for _ in range(30):
            opt.zero_grad()
            terms = diff.training_losses(m, x0, None, t=t, model_kwargs={"y": y}, noise=noise)
            loss = terms["loss"].mean()
            loss.backward()
            opt.step()
            out.append(float(loss))
        return out
    a, b, b2 = run("bf16"), run("fp8"), run("fp8")
    print(f"[fp8 training] bf16 {a[0]:.4f} -> {a[-1]:.4f}; fp8 (delayed scaling after the first step) {b[0]:.4f} -> {b[-1]:.4f}")
    assert a[-1] < 0.8 * a[0] and b[-1] < 0.8 * b[0]
    assert max(abs(x - y) / x for x, y in zip(a, b)) < 5e-2
    assert b == b2           # the delayed-scaling state machine (atomic max of |x| bits) is deterministic


def test_dit_fp8_fused_epilogue_quantisation_is_bitwise_the_separate_quantiser():
    """fp8 mode with delayed scaling: writing `a` and d hidden as fp8 from the GELU / GELU' epilogues must reproduce, bit for bit,
    the run that writes them as bf16 and quantises them in a separate pass (same scales, same bytes, same running maxima)."""
    def run(fuse):
        random.seed(3); np.random.seed(3); torch.manual_seed(3)
        m = vaw_amd.DiT(image_size=32, patch_size=4, in_channels=4, hidden_size=768, depth=3, num_heads=12, class_dropout_prob=0.0,
                        num_classes=10, compute_dtype="fp8")
        m.fp8_fuse_epilogue = fuse
        perturb_(m, 7, std=0.02)
        m = m.to(DEV).train()
        opt = vaw_amd.FusedAdamW(m, lr=1e-3, betas=(0.9, 0.95), weight_decay=0.0, eps=1e-8)
        diff = _diffusion(_args(dict(kind="dit", classes=10, size=32)))
        g = torch.Generator().manual_seed(1)
        x0 = (torch.randn(16, 4, 32, 32, generator=g) * 0.5).to(DEV)
        noise = torch.randn(16, 4, 32, 32, generator=g).to(DEV)
        t = torch.randint(0, 1000, (16,), generator=g).to(DEV)
        y = torch.randint(0, 10, (16,), generator=g).to(DEV)
        losses = []
        for _ in range(4):                  # step 1 measures the scales just in time; steps 2-4 run delayed (and fused)
            opt.zero_grad()
            terms = diff.training_losses(m, x0, None, t=t, model_kwargs={"y": y}, noise=noise)
            terms["loss"].mean().backward()
            opt.step()
            losses.append(terms["mse"].detach().cpu())
        return torch.stack(losses), m._flat.detach().cpu().clone()
    (la, pa), (lb, pb) = run(True), run(False)
    assert torch.equal(la, lb) and torch.equal(pa, pb)


def test_full_batch_properties_fp8():
    """BASELINE config 5 at its per-GPU batch (DiT-XL/2, 128 images, fp8 GEMMs) through the DELAYED-scaling recipe bench.py times:
    one just-in-time calibration pass, then >= 3 passes on last pass's scales (quantisers, fused fp8 epilogues and row kernels,
    grouped fp8 weight gradients).  No oracle at this size: determinism (two models stepped identically end bitwise equal),
    per-sample independence of the objective, finite non-zero gradients, and the loss of pass 4 within 2 % of the bf16 model's
    on the same inputs (the drift of fp8 at batch 2 against the reference fixture is test_dit_xl2_fp8_drift_vs_reference's)."""
    c = CONFIGS["dit_xl2"]
    B, S = c["full_batch"], c["size"]
    diff = _diffusion(_args(c))
    g = torch.Generator().manual_seed(3)
    x0 = (torch.randn(B, 4, S, S, generator=g) * 0.5).to(DEV)
    noise = torch.randn(B, 4, S, S, generator=g).to(DEV)
    t = torch.randint(0, 1000, (B,), generator=g).to(DEV)
    y = torch.randint(0, 1000, (B,), generator=g).to(DEV)

    def passes(dtype, n):
        m = _build(c, dtype).to(DEV).train()
        m.ensure_flat()
        out = []
        for _ in range(n):
            m.zero_grad_flat()
            terms = diff.training_losses(m, x0, None, t=t, model_kwargs={"y": y}, noise=noise)
            terms["loss"].mean().backward()
            out.append((terms["mse"].detach().clone(), m.flat_grads().clone()))
        return m, out

    m1, r1 = passes("fp8", 4)
    ws = m1._ws_cur
    assert ws.fp8 and ws.d_fwd and ws.d_bwd, "passes 2.. must run on delayed scales"
    _, r2 = passes("fp8", 4)
    for (a_mse, a_g), (b_mse, b_g) in zip(r1, r2):
        assert torch.equal(a_mse, b_mse) and torch.equal(a_g, b_g)                     # bitwise reproducible, every pass
    mse4, g4 = r1[-1]
    assert torch.isfinite(mse4).all() and torch.isfinite(g4).all() and float(g4.abs().max()) > 0
    # weights did not move: passes 2-4 differ only by their scales (pass 1: exact amax, later: previous amax x margin)
    for mse_k, _ in r1[1:]:
        torch.testing.assert_close(mse_k, r1[0][0], rtol=3e-2, atol=1e-4)
    idx = torch.tensor([1, B - 3], device=DEV)
    sub = diff.training_losses(m1, x0[idx], None, t=t[idx], model_kwargs={"y": y[idx]}, noise=noise[idx])     # (new workspace: just-in-time)
    torch.testing.assert_close(sub["mse"], mse4[idx], rtol=3e-2, atol=1e-4)
    _, rb = passes("bf16", 1)
    rel = float(((mse4 - rb[0][0]).abs() / rb[0][0].abs()).max())
    print(f"[fp8 delayed scaling, batch {B}] per-sample mse vs bf16: max rel {rel:.3e}")
    assert rel < 2e-2


def test_baseline_configs_run_without_fallback_kernels():
    """One bf16 training pass of each BASELINE model at batch 2 (and the CIFAR-shaped UNet of config 1 at 16) must stay on the
    implicit-GEMM / MFMA kernels: vaw_amd.ops.fallbacks counts every shape that dropped to im2col + GEMM."""
    from vaw_amd import ops
    ops.fallbacks.clear()
    cfgs = dict(CONFIGS)
    cfgs["dit_b4"] = dict(kind="dit", make=lambda dt: vaw_amd.DiT_B(image_size=32, patch_size=4, in_channels=4, class_dropout_prob=0.0,
                                                                    num_classes=1000, learn_sigma=False, compute_dtype=dt),
                          size=32, chans=8, classes=1000, full_batch=256)
    cfgs["unet32"] = dict(kind="unet", make=lambda dt: vaw_amd.UNetModel(32, 3, 64, 3, 2, attention_resolutions=(), channel_mult=(1, 2, 2, 2),
                                                                         num_heads=4, use_scale_shift_norm=True, resblock_updown=True,
                                                                         use_new_attention_order=True, compute_dtype=dt),
                          size=32, chans=3, classes=0, full_batch=16)
    for name, c in cfgs.items():
        B = 16 if name == "unet32" else 2
        m = _build(c, "bf16").to(DEV).train()
        diff = _diffusion(_args(c))
        g = torch.Generator().manual_seed(2)
        Cx = 4 if c["kind"] == "dit" else 3
        x0 = (torch.randn(B, Cx, c["size"], c["size"], generator=g) * 0.5).to(DEV)
        kw = {"y": torch.randint(0, 1000, (B,), generator=g).to(DEV)} if c["classes"] else {}
        terms = diff.training_losses(m, x0, None, model_kwargs=kw)
        terms["loss"].mean().backward()
        torch.cuda.synchronize()
        assert ops.fallbacks == {}, (name, ops.fallbacks)
        del m
