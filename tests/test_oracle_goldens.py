"""Pin the CPU oracle against fixtures generated from the unmodified reference
(tests/golden/make_goldens.py).  CPU only."""
import copy
import os
import random

import numpy as np
import pytest
import torch

from conftest import (GOLDEN, Pbar, assert_fingerprints, base_args, fake_model, load_json, load_pt, perturb_,
                      synth_loader)
from oracle import diffusion as od
from oracle import dit as odit
from oracle import resample as ores
from oracle import trainer as otr
from oracle import unet as ounet


def _diffusion(sched="cosine", mt="EPSILON", wt="lambda", T=1000):
    return od.GaussianDiffusion(args=base_args(weight_type=wt), betas=od.get_named_beta_schedule(sched, T),
                                model_mean_type=od.ModelMeanType[mt], model_var_type=od.ModelVarType.FIXED_LARGE,
                                loss_type=od.LossType.MSE, rescale_timesteps=True)


def test_tables_bit_exact():
    g = np.load(os.path.join(GOLDEN, "diffusion_tables.npz"))
    for sched in ("linear", "cosine", "linear_logsnr"):
        d = _diffusion(sched)
        for k in g.files:
            if k.startswith(sched + "."):
                np.testing.assert_array_equal(getattr(d, k.split(".", 1)[1]), g[k], err_msg=k)
    np.testing.assert_array_equal(od.get_named_beta_schedule("linear", 250), g["linear250.betas"])
    np.testing.assert_array_equal(od.get_named_beta_schedule("cosine", 50), g["cosine50.betas"])
    with pytest.raises(NotImplementedError):
        od.get_named_beta_schedule("nope", 10)


def test_loss_weights_all_types():
    rec = load_json("loss_weight.json")
    d = _diffusion()
    t = torch.tensor(rec["t"])
    for key, exp in rec["diffusion"].items():
        mt, wt = key.split("/")
        a = od.extract(d.sqrt_alphas_cumprod, t, t.shape).clone()
        s = od.extract(d.sqrt_one_minus_alphas_cumprod, t, t.shape).clone()
        if "error" in exp:
            with pytest.raises(ValueError):
                od.compute_mse_loss_weight(od.ModelMeanType[mt], wt, t, a, s, 1, 1)
            continue
        w = od.compute_mse_loss_weight(od.ModelMeanType[mt], wt, t, a, s, 1, 1)
        assert str(w.dtype) == exp["dtype"], key
        np.testing.assert_array_equal(w.double().numpy(), np.array(exp["w"]), err_msg=key)
    tf = torch.tensor(rec["flow_t"], dtype=torch.float32)
    for key, exp in rec["flow"].items():
        parts = key.split("/")
        fm = od.FlowMatching(args=base_args(path_type=parts[0]), model_mean_type=od.ModelMeanType.VECTOR)
        a, s, da, ds = fm.interpolant(tf)
        if parts[1] == "interpolant":
            for n, v in zip(("a", "s", "da", "ds"), (a, s, da, ds)):
                np.testing.assert_array_equal(v.double().numpy(), np.array(exp[n]), err_msg=key + n)
            continue
        if "error" in exp:
            with pytest.raises(ValueError):
                od.compute_mse_loss_weight(od.ModelMeanType[parts[1]], parts[2], tf, a.clone(), s.clone(), 1, 1)
            continue
        w = od.compute_mse_loss_weight(od.ModelMeanType[parts[1]], parts[2], tf, a.clone(), s.clone(), 1, 1)
        assert str(w.dtype) == exp["dtype"], key
        np.testing.assert_array_equal(w.double().numpy(), np.array(exp["w"]), err_msg=key)
    alpha, sigma = torch.tensor([0.0, 0.5]), torch.tensor([1.0, 0.8])
    w = od.compute_mse_loss_weight(od.ModelMeanType.EPSILON, "lambda", torch.tensor([0, 1]), alpha, sigma)
    assert w.tolist() == pytest.approx(rec["edge"]["alias"]["w"])
    assert sigma.tolist() == pytest.approx(rec["edge"]["alias"]["sigma_after"])   # in-place alias reproduced


def test_objective_bit_exact():
    g = load_pt("objective.pt")
    x0, noise, t, y = g["x0"], g["noise"], g["t"], g["y"]
    for sched in ("cosine", "linear"):
        for mt in ("EPSILON", "START_X", "VELOCITY"):
            for wt in ("lambda", "constant", "min_snr_5"):
                d = _diffusion(sched, mt, wt)
                terms = d.training_losses(fake_model, x0, None, t=t, model_kwargs={"y": y}, noise=noise)
                assert torch.equal(terms["mse"].float(), g[f"{sched}/{mt}/{wt}/mse"]), (sched, mt, wt)
                assert torch.equal(terms["loss"].float(), g[f"{sched}/{mt}/{wt}/loss"])
            d = _diffusion(sched, mt)
            assert torch.equal(d.q_sample(x0, t, noise), g[f"{sched}/{mt}/x_t"])
            assert torch.equal(d.compute_target(x0, noise, t), g[f"{sched}/{mt}/target"])
    torch.manual_seed(42)
    assert torch.equal(_diffusion().training_losses(fake_model, x0, None)["mse"], g["seed42/mse"])
    for path in ("linear", "cosine", "linear_logsnr"):
        for mt in ("VECTOR", "EPSILON", "VELOCITY", "START_X"):
            fm = od.FlowMatching(args=base_args(path_type=path), model_mean_type=od.ModelMeanType[mt])
            terms = fm.training_losses(fake_model, x0, None, t=g["flow/t"], model_kwargs={"y": y}, noise=noise)
            assert torch.equal(terms["mse"].float(), g[f"flow/{path}/{mt}/mse"]), (path, mt)
    fm = od.FlowMatching(args=base_args(path_type="linear", time_dist=["lognorm", -0.8, 0.8]),
                         model_mean_type=od.ModelMeanType.VECTOR)
    torch.manual_seed(5)
    assert torch.equal(fm.sample_t(x0), g["flow/lognorm_t"])


VB_CASES = [(sched, mt, vt, lt) for sched in ("cosine", "linear") for mt in ("EPSILON", "START_X")
            for vt in ("LEARNED_RANGE", "LEARNED", "FIXED_LARGE", "FIXED_SMALL") for lt in ("MSE", "RESCALED_MSE", "KL", "RESCALED_KL")
            if vt.startswith("LEARNED") or lt in ("KL", "RESCALED_KL")]


def test_variational_bound_objectives_vs_reference():
    """Learned-variance vb term and the pure KL losses (reference :775-808, :865-906, tools/losses.py) incl. the
    decoder-NLL branch at t = 0 and d loss / d model_output; the oracle's op order is the reference's, so bit-exact."""
    g = load_pt("vb_objective.pt")
    x0, noise, t = g["x0"], g["noise"], g["t"]
    for sched, mt, vt, lt in VB_CASES:
        learned = vt.startswith("LEARNED")
        d = od.GaussianDiffusion(args=base_args(weight_type="lambda", learn_sigma=learned),
                                 betas=od.get_named_beta_schedule(sched, 1000), model_mean_type=od.ModelMeanType[mt],
                                 model_var_type=od.ModelVarType[vt], loss_type=od.LossType[lt], rescale_timesteps=True)
        P = (g["P"] if learned else g["P"][:, :3]).clone().requires_grad_(True)
        terms = d.training_losses(lambda x, ts, **kw: P, x0, None, t=t, noise=noise)
        terms["loss"].sum().backward()
        key = f"{sched}/{mt}/{vt}/{lt}"
        for k, v in terms.items():
            assert torch.equal(v.detach().float(), g[f"{key}/{k}"]), (key, k)
        torch.testing.assert_close(P.grad, g[f"{key}/dP"], rtol=1e-6, atol=1e-9, msg=key)
    d = od.GaussianDiffusion(args=base_args(learn_sigma=True), betas=od.get_named_beta_schedule("cosine", 1000),
                             model_mean_type=od.ModelMeanType.VELOCITY, model_var_type=od.ModelVarType.LEARNED_RANGE,
                             loss_type=od.LossType.MSE, rescale_timesteps=True)
    with pytest.raises(RuntimeError):            # the reference cannot broadcast here either (:394-399)
        d.training_losses(lambda x, ts, **kw: g["P"], x0, None, t=t, noise=noise)


def _fwd_bwd(m, x, t, y, gout):
    x = x.clone().requires_grad_(True)
    m.zero_grad()
    raw = m(x, t, y=y) if y is not None else m(x, t)
    out = raw[0] if isinstance(raw, tuple) else raw
    (out * gout).sum().backward()
    return out.detach(), x.grad, {k: p.grad for k, p in m.named_parameters() if p.grad is not None}


@pytest.mark.parametrize("tag", ["p2", "p4"])
def test_dit_tiny_matches_reference(tag):
    g = load_pt("dit_tiny.pt")
    torch.manual_seed(11)
    m = odit.DiT(in_channels=4, class_dropout_prob=0.0, num_classes=10, learn_sigma=False, **g[f"{tag}/kw"])
    assert_fingerprints(m.state_dict(), g[f"{tag}/init_sd"], 1e-6, 1e-9, "init")
    m.train()
    out0, aux = m(g[f"{tag}/x"], g[f"{tag}/t"], g[f"{tag}/y"])
    assert aux is None and float(out0.abs().max()) == 0.0 == float(g[f"{tag}/out_at_init_absmax"])  # adaLN-Zero
    perturb_(m, 99)
    assert_fingerprints(m.state_dict(), g[f"{tag}/sd"], 1e-6, 1e-9, "perturbed")
    out, gx, grads = _fwd_bwd(m, g[f"{tag}/x"], g[f"{tag}/t"], g[f"{tag}/y"], g[f"{tag}/gout"])
    torch.testing.assert_close(out, g[f"{tag}/out"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(gx, g[f"{tag}/gx"], rtol=1e-5, atol=1e-6)
    assert_fingerprints(grads, g[f"{tag}/grads"], 1e-4, 1e-6, "grads")
    # unpatchify o patchify == identity (needs no timm)
    p = m.x_embedder.patch_size[0]
    x = g[f"{tag}/x"]
    n, c, H, W = x.shape
    tok = x.reshape(n, c, H // p, p, W // p, p).permute(0, 2, 4, 3, 5, 1).reshape(n, (H // p) * (W // p), p * p * c)
    assert torch.equal(m.unpatchify(tok), x)


@pytest.mark.parametrize("tag", ["new", "legacy", "legacy_ss"])
def test_unet_tiny_matches_reference(tag):
    g = load_pt("unet_tiny.pt")
    torch.manual_seed(21)
    m = ounet.UNetModel(**g[f"{tag}/kw"])
    assert_fingerprints(m.state_dict(), g[f"{tag}/init_sd"], 1e-6, 1e-9, "init")
    m.train()
    perturb_(m, 77, std=0.03)
    y = g[f"{tag}/y"] if g[f"{tag}/y"].numel() else None
    out, gx, grads = _fwd_bwd(m, g[f"{tag}/x"], g[f"{tag}/t"], y, g[f"{tag}/gout"])
    torch.testing.assert_close(out, g[f"{tag}/out"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(gx, g[f"{tag}/gx"], rtol=1e-5, atol=1e-6)
    assert_fingerprints(grads, g[f"{tag}/grads"], 1e-4, 1e-6, "grads")
    torch.testing.assert_close(ounet.timestep_embedding(g["temb/t"], 64), g["temb/out64"], rtol=0, atol=0)
    torch.testing.assert_close(ounet.timestep_embedding(g["temb/t"], 33), g["temb/out33"], rtol=0, atol=0)


def test_unet_dropout_oracle_vs_reference():
    """nn.Dropout inside the ResBlocks: the oracle under the reference's seed reproduces its masks, hence outputs and gradients."""
    g = load_pt("unet_dropout.pt")
    torch.manual_seed(21)
    m = ounet.UNetModel(**g["kw"])
    m.train()
    perturb_(m, 77, std=0.03)
    torch.manual_seed(5)
    out, gx, grads = _fwd_bwd(m, g["x"], g["t"], g["y"], g["gout"])
    torch.testing.assert_close(out, g["out"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(gx, g["gx"], rtol=1e-5, atol=1e-6)
    assert_fingerprints(grads, g["grads"], 1e-4, 1e-6, "grads")
    m.eval()
    with torch.no_grad():
        torch.testing.assert_close(m(g["x"], g["t"], y=g["y"]), g["out_eval"], rtol=1e-5, atol=1e-6)


def test_unet_factory_param_counts():
    g = load_pt("unet_tiny.pt")
    assert sum(p.numel() for p in ounet.UNet_64(class_cond=False).parameters()) == g["nparams/UNet_64_uncond"]
    assert sum(p.numel() for p in ounet.ADM_64(num_classes=1000, class_cond=True).parameters()) == g["nparams/ADM_64_c1000"]


@pytest.mark.parametrize("name", ["unet64", "adm64", "dit_xl2"])
def test_full_size_oracle_models_vs_reference_objective(name):
    """BASELINE configs 2, 3, 5 at full model size, batch 2 (tests/golden/bigcfg.pt, written from the unmodified reference):
    the oracle's seed-reconstructed weights and its per-sample terms['mse'] (forward only: the CPU suite stays short; the
    gradients and Trainer steps of the same fixture are checked on the GPU path, tests/test_gpu_bigcfg.py)."""
    g = load_pt("bigcfg.pt")[name]
    random.seed(42); np.random.seed(42); torch.manual_seed(42)
    if name == "unet64":
        m, args = ounet.UNet_64(class_cond=False), base_args(image_size=64, dataset="ImageNet")
    elif name == "adm64":
        m, args = ounet.ADM_64(num_classes=1000, class_cond=True), base_args(image_size=64, dataset="ImageNet", class_cond=True)
    else:
        m = odit.DiT_XL(image_size=32, patch_size=2, in_channels=4, class_dropout_prob=0.0, num_classes=1000, learn_sigma=False)
        args = base_args(in_chans=4, class_cond=True, dataset="Latent", image_size=32)
    perturb_(m, 7, std=0.02)
    assert sum(p.numel() for p in m.parameters()) == g["n_params"]
    assert_fingerprints({k: v.detach() for k, v in m.named_parameters()}, g["params"], 1e-6, 1e-9, "seed-reconstructed weights")
    diff = od.GaussianDiffusion(args=args, betas=od.get_named_beta_schedule("cosine", 1000), model_mean_type=od.ModelMeanType.EPSILON,
                                model_var_type=od.ModelVarType.FIXED_LARGE, loss_type=od.LossType.MSE, rescale_timesteps=True)
    kw = {"y": g["y"]} if g["y"].numel() else {}
    with torch.no_grad():
        terms = diff.training_losses(m, g["x"], None, t=g["t"], model_kwargs=kw, noise=g["noise"])
    torch.testing.assert_close(terms["mse"].double(), g["mse"], rtol=2e-6, atol=0)


def _run_trainer(make_model, args, batches, steps, betas2=(0.9, 0.95), var_type="FIXED_LARGE"):
    random.seed(42); np.random.seed(42); torch.manual_seed(42)
    model = make_model()
    ema_model = copy.deepcopy(model)
    opt = torch.optim.AdamW(model.parameters(), lr=args.lr, betas=betas2, weight_decay=0.0, eps=1e-8)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=otr.get_lr_lambda(args))
    diff = od.GaussianDiffusion(args=args, betas=od.get_named_beta_schedule(args.path_type, 1000),
                                model_mean_type=od.ModelMeanType.EPSILON, model_var_type=od.ModelVarType[var_type],
                                loss_type=od.LossType.MSE, rescale_timesteps=True)
    tr = otr.Trainer(args, torch.device("cpu"), model, ema_model, opt, sched, diff, batches, Pbar())
    losses = [tr.train_step(s) for s in range(1, steps + 1)]
    psum = float(sum(p.double().abs().sum() for p in model.parameters()))
    esum = float(sum(v.double().abs().sum() for v in ema_model.state_dict().values()))
    return losses, psum, esum, sched.get_last_lr()[0]


CFG1 = lambda: ounet.UNetModel(32, 3, 64, 3, 2, attention_resolutions=(), channel_mult=(1, 2, 2, 2), num_heads=4,
                               use_scale_shift_norm=True, resblock_updown=True, use_new_attention_order=True)
TINY_DIT = lambda: odit.DiT(image_size=8, patch_size=2, in_channels=4, hidden_size=64, depth=2, num_heads=2,
                            class_dropout_prob=0.0, num_classes=10, learn_sigma=False)


@pytest.mark.parametrize("name,make,args,loader,steps", [
    ("cfg1", CFG1, base_args(), lambda: synth_loader(16, 3, 32, 4, 0), 5),
    ("cfg1_accum2_clip", CFG1, base_args(grad_accumulation=2, grad_clip=1.0), lambda: synth_loader(16, 3, 32, 4, 0), 3),
    ("cfg1_warmup_cosine_minsnr", CFG1, base_args(weight_type="min_snr_5", warmup_steps=2, cosine_decay=True,
                                                   total_steps=10, final_lr=1e-5), lambda: synth_loader(16, 3, 32, 4, 0), 4),
    ("dit_tiny_latent", TINY_DIT, base_args(in_chans=4, class_cond=True, dataset="Latent", image_size=8, lr=1e-3),
     lambda: synth_loader(8, 8, 8, 3, 10, latent=True), 6),
])
def test_trainer_trajectories(name, make, args, loader, steps):
    exp = load_json("trainer.json")[name]
    losses, psum, esum, lr = _run_trainer(make, args, loader(), steps)
    np.testing.assert_allclose(losses, exp["losses"], rtol=2e-6)
    assert psum == pytest.approx(exp["param_abs_sum"], rel=1e-7)
    assert esum == pytest.approx(exp["ema_abs_sum"], rel=1e-9)
    assert lr == pytest.approx(exp["lr_last"], rel=1e-12)


TINY_DIT_LS = lambda: odit.DiT(image_size=8, patch_size=2, in_channels=4, hidden_size=64, depth=2, num_heads=2,
                               class_dropout_prob=0.0, num_classes=10, learn_sigma=True)
TINY_UNET_LS = lambda: ounet.UNetModel(16, 3, 32, 6, 1, attention_resolutions=(2,), channel_mult=(1, 2), num_heads=2,
                                       use_scale_shift_norm=True, resblock_updown=True, use_new_attention_order=True)


@pytest.mark.parametrize("name,make,args,loader,steps", [
    ("dit_tiny_learn_sigma", TINY_DIT_LS, base_args(in_chans=4, class_cond=True, dataset="Latent", image_size=8, lr=1e-3,
                                                    learn_sigma=True), lambda: synth_loader(8, 8, 8, 3, 10, latent=True), 6),
    ("unet_tiny_learn_sigma", TINY_UNET_LS, base_args(image_size=16, lr=1e-3, learn_sigma=True),
     lambda: synth_loader(8, 3, 16, 3, 0), 5),
])
def test_trainer_trajectories_learned_variance(name, make, args, loader, steps):
    """loss = mse + vb with a 2C-channel model output (LEARNED_RANGE), reference Trainer trajectories."""
    exp = load_json("trainer_vb.json")[name]
    losses, psum, esum, lr = _run_trainer(make, args, loader(), steps, var_type="LEARNED_RANGE")
    np.testing.assert_allclose(losses, exp["losses"], rtol=2e-6)
    assert psum == pytest.approx(exp["param_abs_sum"], rel=1e-7)
    assert esum == pytest.approx(exp["ema_abs_sum"], rel=1e-9)


def test_sampling_side_vs_reference():
    """space_timesteps, SpacedDiffusion + p_sample / ddim_sample loops under the CPU RNG stream, IntervalCFG
    (tests/golden/sampling.pt, written by the unmodified reference)."""
    from conftest import SAMPLING_CASES, sampling_model, sampling_model_2c
    from oracle import respace as orsp, sampler as osam
    g = load_pt("sampling.pt")
    for k, v in g["space"].items():
        assert sorted(orsp.space_timesteps(1000, k)) == v, k
    assert sorted(orsp.space_timesteps(300, [10, 15, 20])) == g["space_300_10_15_20"]
    with pytest.raises(ValueError):
        orsp.space_timesteps(1000, "ddim999")
    shape, y = (3, 3, 8, 8), torch.tensor([1, 5, 9])
    for name, sched, mt, vt, respacing, kind, eta, clip in SAMPLING_CASES:
        learned = vt.startswith("LEARNED")
        d = orsp.SpacedDiffusion(use_timesteps=orsp.space_timesteps(1000, respacing), args=base_args(learn_sigma=learned),
                                 betas=od.get_named_beta_schedule(sched, 1000), model_mean_type=od.ModelMeanType[mt],
                                 model_var_type=od.ModelVarType[vt], loss_type=od.LossType.MSE, rescale_timesteps=True)
        exp = g["loops"][name]
        assert d.timestep_map == exp["timestep_map"].tolist()
        torch.manual_seed(123)
        kw = dict(clip_denoised=clip, model_kwargs={"y": y})
        loop = d.ddim_sample_loop_progressive(sampling_model_2c if learned else sampling_model, shape, eta=eta, **kw) \
            if kind == "ddim" else d.p_sample_loop_progressive(sampling_model_2c if learned else sampling_model, shape, **kw)
        traj = [o["sample"] for o in loop]
        assert len(traj) == exp["n"]
        for k, v in (("first", traj[0]), ("mid", traj[len(traj) // 2]), ("final", traj[-1])):
            assert torch.equal(v, exp[k]), (name, k)
    x, yy = g["cfg_x"], g["cfg_y"]
    for nm, scale, interval, tval in [("plain", 1.0, (-1.0, -1.0), 500.0), ("always", 2.5, (-1.0, -1.0), 500.0),
                                      ("inside", 1.8, (100.0, 600.0), 300.0), ("outside", 1.8, (100.0, 600.0), 800.0)]:
        m = osam.IntervalCFG(sampling_model, 10, scale, interval, True)
        assert torch.equal(m(x, torch.full((4,), tval), y=yy), g["cfg"][nm]), nm


def test_misc_lr_resampler_latent():
    rec = load_json("misc.json")
    for s, a, b, c in rec["lr"]:
        assert otr.warmup_cosine_lr(s, 5, 50, 1e-4, 1e-6, True) == a
        assert otr.warmup_cosine_lr(s, 5, 50, 1e-4, 1e-6, False) == b
        assert otr.warmup_cosine_lr(s, 0, 50, 1e-4, 0.0, True) == c
    from types import SimpleNamespace
    diff = SimpleNamespace(num_timesteps=20)
    s = ores.create_named_schedule_sampler("loss-second-moment", diff)
    assert s.weights().tolist() == rec["lsm_weights_before"]
    rng = np.random.RandomState(0)
    for _ in range(15):
        s.update_with_all_losses(list(range(20)), (rng.rand(20) * (1 + np.arange(20))).tolist())
    assert s.weights().tolist() == rec["lsm_weights_after"]
    np.random.seed(3)
    idx, w = s.sample(16, "cpu")
    assert idx.tolist() == rec["lsm_sample_idx"] and w.double().tolist() == rec["lsm_sample_w"]
    u = ores.create_named_schedule_sampler("uniform", diff)
    np.random.seed(3)
    idx, w = u.sample(8, "cpu")
    assert idx.tolist() == rec["uni_sample_idx"] and w.double().tolist() == rec["uni_sample_w"]
    with pytest.raises(NotImplementedError):
        ores.create_named_schedule_sampler("nope", diff)
    lat = torch.tensor(rec["sfl_in"], dtype=torch.float32)
    torch.manual_seed(1)
    assert otr.sample_from_latent(lat, 0.18215).double().tolist() == rec["sfl_out"]


@pytest.mark.parametrize("key", ["h4_d64", "h6_d96"])
def test_timm_restatement_vs_reference_uvit_attention_and_mlp(key):
    """The stand-in for timm==0.9.2's Attention / Mlp (oracle/timm_restatement.py; timm is not importable here) against the
    reference's OWN statements of the same arithmetic: models/uvit.py:55-93 `Attention` ('math' and 'flash' modes) and
    tools/timm.py:96-112 `Mlp`, run unmodified by tests/golden/make_goldens.py::gen_uvit_anchor.  Same state_dict keys, same
    weights -> same output and input gradient: a reference-held pin of the (K H D) qkv packing, the hd^-1/2 scale and the
    head merge.  What stays unpinned is only "timm 0.9.2 == the reference's uvit statement"."""
    from oracle import timm_restatement as tr
    g = load_pt("uvit_anchor.pt")[key]
    heads, dim = int(key[1]), int(key.split("_d")[1])
    att = tr.Attention(dim, num_heads=heads, qkv_bias=True)
    att.load_state_dict(g["attn_state"], strict=True)
    mlp = tr.Mlp(dim, 4 * dim, act_layer=lambda: torch.nn.GELU(approximate="tanh"))
    mlp.load_state_dict(g["mlp_state"], strict=True)
    for mod, names in ((att, ("attn_math", "attn_flash")), (mlp, ("mlp",))):
        x = g["x"].clone().requires_grad_(True)
        y = mod(x)
        (gx,) = torch.autograd.grad(y, x, g["gy"])
        for n in names:
            torch.testing.assert_close(y.detach(), g[n + "_y"], rtol=2e-6, atol=2e-6)
            torch.testing.assert_close(gx, g[n + "_gx"], rtol=2e-6, atol=2e-6)
