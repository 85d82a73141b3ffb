#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by running the UNMODIFIED
reference (/root/reference) on CPU in the build container.

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/make_goldens.py

The reference imports third-party modules that are absent here (torchdiffeq,
torchvision, diffusers, timm, ...).  None of them is touched by
`training_losses` / `train_step` with learn_align=False, so they are replaced
by name-only stub modules.  The single exception is
`timm.models.vision_transformer.{Attention,Mlp,PatchEmbed}`, whose arithmetic
DiT needs: it is served by oracle/timm_restatement.py (parity unpinned at that
boundary; everything the reference itself owns in DiT is pinned by these files).

Only data is written: inputs, seeds, weights of tiny models, expected outputs.
The reference's source never enters the repository.  Fixtures are loaded in
tests with torch.load(weights_only=True) / numpy / json.
"""
import json
import os
import sys
import types
import warnings
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
warnings.filterwarnings("ignore")


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_stubs():
    sys.path.insert(0, REPO)
    from oracle import timm_restatement as tr
    _stub("torchdiffeq", odeint=None)
    _stub("torchvision")
    _stub("torchvision.utils", make_grid=None, save_image=None)
    _stub("torchvision.transforms")
    _stub("diffusers")
    _stub("diffusers.models", AutoencoderKL=None)
    _stub("timm")
    _stub("timm.data", IMAGENET_DEFAULT_MEAN=None, IMAGENET_DEFAULT_STD=None)
    _stub("timm.models")
    _stub("timm.models.vision_transformer", Attention=tr.Attention, Mlp=tr.Mlp, PatchEmbed=tr.PatchEmbed)
    _stub("tools.encoders", load_encoders=None)
    sys.path.insert(0, REF)


def base_args(**kw):
    a = dict(weight_type="lambda", gamma=0.0, learn_sigma=False, p2_gamma=1, p2_k=1, time_dist=["uniform", -0.8, 0.8],
             learn_align=False, align_type="mse", amp=False, dataset="CIFAR-10", class_cond=False, parallel=False,
             grad_accumulation=1, in_chans=3, latent_scale=0.18215, grad_clip=None, ema_decay=0.9999,
             enc_type="dinov2-vit-b", image_size=32, path_type="cosine", sampler_type="ode", lr=1e-4, final_lr=0.0,
             warmup_steps=0, total_steps=1000, cosine_decay=False)
    a.update(kw)
    return SimpleNamespace(**a)


class Pbar:
    def update(self, n):
        pass

    def set_postfix(self, **kw):
        pass


def perturb_(model, seed, std=0.05):
    """Make zero-initialised layers non-trivial, deterministically."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in model.parameters():
            if p.requires_grad:
                p.add_(torch.randn(p.shape, generator=g) * std)


def tolist(t):
    return t.detach().double().cpu().numpy().tolist()


def summarize(tensors):
    """Per-tensor fingerprint: [sum, abs-sum, l2] in float64 plus a 64-element strided sample.
    Full weights are NOT stored: tests rebuild the model from the same seed with the oracle's
    constructor (which mirrors the reference's RNG order) and must reproduce these fingerprints."""
    out = {}
    for k, v in tensors.items():
        f = v.detach().double().flatten()
        stride = max(1, f.numel() // 64)
        out[k] = {"stats": torch.stack([f.sum(), f.abs().sum(), f.norm()]), "sample": f[::stride][:64].clone()}
    return out


def gen_tables(gd):
    out = {}
    for name in ("linear", "cosine", "linear_logsnr"):
        betas = gd.get_named_beta_schedule(name, 1000)
        d = gd.GaussianDiffusion(args=base_args(), betas=betas, model_mean_type=gd.ModelMeanType.EPSILON,
                                 model_var_type=gd.ModelVarType.FIXED_LARGE, loss_type=gd.LossType.MSE,
                                 rescale_timesteps=True, device="cpu")
        for k in ("betas", "alphas_cumprod", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod",
                  "log_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod",
                  "posterior_variance", "posterior_log_variance_clipped", "posterior_mean_coef1",
                  "posterior_mean_coef2"):
            out[f"{name}.{k}"] = getattr(d, k)
    out["linear250.betas"] = gd.get_named_beta_schedule("linear", 250)
    out["cosine50.betas"] = gd.get_named_beta_schedule("cosine", 50)
    np.savez_compressed(os.path.join(HERE, "diffusion_tables.npz"), **out)


WEIGHT_TYPES = ["constant", "lambda", "min_snr_5", "min_snr_1.5", "max_snr_2", "debias", "p2", "min_debias",
                "max_debias", "trunc_snr", "snr", "inv_snr", "bogus"]


def gen_loss_weights(gd):
    ts = [0, 1, 10, 250, 500, 750, 998, 999]
    rec = {"t": ts, "diffusion": {}, "flow": {}, "edge": {}}
    betas = gd.get_named_beta_schedule("cosine", 1000)
    d = gd.GaussianDiffusion(args=base_args(), betas=betas, model_mean_type=gd.ModelMeanType.EPSILON,
                             model_var_type=gd.ModelVarType.FIXED_LARGE, loss_type=gd.LossType.MSE,
                             rescale_timesteps=True, device="cpu")
    t = torch.tensor(ts)
    for mt in ("EPSILON", "START_X", "VELOCITY", "VECTOR"):
        for wt in WEIGHT_TYPES:
            alpha = gd._extract_into_tensor(d.sqrt_alphas_cumprod, t, t.shape).clone()
            sigma = gd._extract_into_tensor(d.sqrt_one_minus_alphas_cumprod, t, t.shape).clone()
            try:
                w = gd.compute_mse_loss_weight(gd.ModelMeanType[mt], wt, t, alpha, sigma, 1, 1)
                rec["diffusion"][f"{mt}/{wt}"] = {"w": tolist(w), "dtype": str(w.dtype)}
            except ValueError as e:
                rec["diffusion"][f"{mt}/{wt}"] = {"error": "ValueError"}
    # continuous-time (flow) coefficients incl. the snr==0 edge (t=1 on the linear path => alpha=0)
    tf = torch.tensor([0.0, 0.001, 0.25, 0.5, 0.9, 1.0])
    rec["flow_t"] = tolist(tf)
    for path in ("linear", "cosine", "linear_logsnr"):
        fm = gd.FlowMatching(args=base_args(path_type=path), model_mean_type=gd.ModelMeanType.VECTOR)
        a, s, da, ds = fm.interpolant(tf)
        rec["flow"][f"{path}/interpolant"] = {"a": tolist(a), "s": tolist(s), "da": tolist(da), "ds": tolist(ds)}
        for mt in ("EPSILON", "START_X", "VELOCITY", "VECTOR"):
            for wt in ("constant", "lambda", "min_snr_5"):
                a, s, _, _ = fm.interpolant(tf)
                try:
                    w = gd.compute_mse_loss_weight(gd.ModelMeanType[mt], wt, tf, a.clone(), s.clone(), 1, 1)
                    rec["flow"][f"{path}/{mt}/{wt}"] = {"w": tolist(w), "dtype": str(w.dtype)}
                except ValueError:
                    rec["flow"][f"{path}/{mt}/{wt}"] = {"error": "ValueError"}
    # aliasing edge: sigma patched in place where snr==0
    alpha = torch.tensor([0.0, 0.5]); sigma = torch.tensor([1.0, 0.8])
    w = gd.compute_mse_loss_weight(gd.ModelMeanType.EPSILON, "lambda", torch.tensor([0, 1]), alpha, sigma)
    rec["edge"]["alias"] = {"w": tolist(w), "sigma_after": tolist(sigma)}
    json.dump(rec, open(os.path.join(HERE, "loss_weight.json"), "w"))


def fake_model(x, t, **kw):
    """A deterministic stand-in denoiser so the objective can be pinned without a network."""
    return 0.5 * x + 1e-3 * t.view(-1, 1, 1, 1).float() + (0.01 * kw["y"].view(-1, 1, 1, 1).float() if "y" in kw else 0)


def gen_objective(gd):
    g = torch.Generator().manual_seed(7)
    x0 = torch.rand(4, 3, 8, 8, generator=g) * 2 - 1
    noise = torch.randn(4, 3, 8, 8, generator=g)
    t = torch.tensor([0, 17, 500, 999])
    y = torch.tensor([1, 2, 3, 4])
    out = {"x0": x0, "noise": noise, "t": t, "y": y}
    for sched in ("cosine", "linear"):
        betas = gd.get_named_beta_schedule(sched, 1000)
        for mt in ("EPSILON", "START_X", "VELOCITY"):
            for wt in ("lambda", "constant", "min_snr_5"):
                d = gd.GaussianDiffusion(args=base_args(weight_type=wt), betas=betas,
                                         model_mean_type=gd.ModelMeanType[mt],
                                         model_var_type=gd.ModelVarType.FIXED_LARGE, loss_type=gd.LossType.MSE,
                                         rescale_timesteps=True, device="cpu")
                terms = d.training_losses(fake_model, x0, None, t=t, model_kwargs={"y": y}, noise=noise)
                out[f"{sched}/{mt}/{wt}/mse"] = terms["mse"].float()
                out[f"{sched}/{mt}/{wt}/loss"] = terms["loss"].float()
                if wt == "lambda":
                    out[f"{sched}/{mt}/x_t"] = d.q_sample(x0, t, noise)
                    out[f"{sched}/{mt}/target"] = d.compute_target(x0, noise, t).clone()
    # RNG order: noise is drawn BEFORE t
    d = gd.GaussianDiffusion(args=base_args(), betas=gd.get_named_beta_schedule("cosine", 1000),
                             model_mean_type=gd.ModelMeanType.EPSILON, model_var_type=gd.ModelVarType.FIXED_LARGE,
                             loss_type=gd.LossType.MSE, rescale_timesteps=True, device="cpu")
    torch.manual_seed(42)
    out["seed42/mse"] = d.training_losses(fake_model, x0, None)["mse"]
    # flow matching
    tf = torch.tensor([0.03, 0.4, 0.77, 0.999])
    for path in ("linear", "cosine", "linear_logsnr"):
        for mt in ("VECTOR", "EPSILON", "VELOCITY", "START_X"):
            fm = gd.FlowMatching(args=base_args(path_type=path), model_mean_type=gd.ModelMeanType[mt])
            terms = fm.training_losses(fake_model, x0, None, t=tf, model_kwargs={"y": y}, noise=noise)
            out[f"flow/{path}/{mt}/mse"] = terms["mse"].float()
    out["flow/t"] = tf
    fm = gd.FlowMatching(args=base_args(path_type="linear", time_dist=["lognorm", -0.8, 0.8]),
                         model_mean_type=gd.ModelMeanType.VECTOR)
    torch.manual_seed(5)
    out["flow/lognorm_t"] = fm.sample_t(x0)
    torch.save(out, os.path.join(HERE, "objective.pt"))


def gen_vb(gd):
    """Learned-variance / variational-bound objectives (reference gaussian_diffusion.py:775-808, 886-906, tools/losses.py):
    the model is a leaf tensor P (so d loss / d output is pinned too).  t = 0 exercises the decoder-NLL branch and
    x0 holds values beyond +-0.999 for the open-ended bins of discretized_gaussian_log_likelihood."""
    g = torch.Generator().manual_seed(11)
    x0 = torch.rand(4, 3, 8, 8, generator=g) * 2 - 1
    x0[0, 0, 0, 0], x0[0, 0, 0, 1], x0[0, 1, 2, 3], x0[0, 2, 5, 5] = -1.0, 1.0, -0.9995, 0.9999
    noise = torch.randn(4, 3, 8, 8, generator=g)
    P6 = torch.cat([torch.randn(4, 3, 8, 8, generator=g) * 0.5, torch.rand(4, 3, 8, 8, generator=g) * 2 - 1], 1)
    t = torch.tensor([0, 17, 500, 999])
    out = {"x0": x0, "noise": noise, "t": t, "P": P6}
    for sched in ("cosine", "linear"):
        betas = gd.get_named_beta_schedule(sched, 1000)
        # VELOCITY is absent: the reference's _predict_xstart_from_v (:394-399) extracts its coefficients with
        # t.shape instead of x_t.shape and fails to broadcast (RuntimeError) for any image batch
        for mt in ("EPSILON", "START_X"):
            for vt in ("LEARNED_RANGE", "LEARNED", "FIXED_LARGE", "FIXED_SMALL"):
                learned = vt.startswith("LEARNED")
                for lt in ("MSE", "RESCALED_MSE", "KL", "RESCALED_KL"):
                    if not learned and lt in ("MSE", "RESCALED_MSE"):
                        continue                       # the plain objective, pinned by objective.pt
                    d = gd.GaussianDiffusion(args=base_args(weight_type="lambda", learn_sigma=learned), betas=betas,
                                             model_mean_type=gd.ModelMeanType[mt], model_var_type=gd.ModelVarType[vt],
                                             loss_type=gd.LossType[lt], rescale_timesteps=True, device="cpu")
                    P = (P6 if learned else P6[:, :3]).clone().requires_grad_(True)
                    terms = d.training_losses(lambda x, ts, **kw: P, x0, None, t=t, noise=noise)
                    terms["loss"].sum().backward()
                    key = f"{sched}/{mt}/{vt}/{lt}"
                    for k, v in terms.items():
                        out[f"{key}/{k}"] = v.detach().float()
                    out[f"{key}/dP"] = P.grad.clone()
    torch.save(out, os.path.join(HERE, "vb_objective.pt"))


def fwd_bwd(model, x, t, y, gout):
    x = x.clone().requires_grad_(True)
    model.zero_grad()
    raw = model(x, t, y=y) if y is not None else model(x, t)
    out = raw[0] if isinstance(raw, tuple) else raw
    (out * gout).sum().backward()
    grads = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    return out.detach(), x.grad.clone(), grads


def gen_dit_tiny():
    from models.dit import DiT
    rec = {}
    for tag, kw in {"p2": dict(image_size=8, patch_size=2, hidden_size=64, depth=2, num_heads=2),
                    "p4": dict(image_size=16, patch_size=4, hidden_size=128, depth=1, num_heads=2)}.items():
        torch.manual_seed(11)
        m = DiT(in_channels=4, class_dropout_prob=0.0, num_classes=10, learn_sigma=False, **kw)
        rec[f"{tag}/init_sd"] = summarize(m.state_dict())
        m.train()
        g = torch.Generator().manual_seed(3)
        x = torch.randn(3, 4, kw["image_size"], kw["image_size"], generator=g)
        t = torch.tensor([0.0, 421.0, 999.0])
        y = torch.tensor([0, 5, 9])
        out0, _ = m(x, t, y)
        rec[f"{tag}/out_at_init_absmax"] = out0.abs().max()
        perturb_(m, 99)
        gout = torch.randn(out0.shape, generator=g)
        out, gx, grads = fwd_bwd(m, x, t, y, gout)
        rec.update({f"{tag}/kw": kw, f"{tag}/sd": summarize(m.state_dict()),
                    f"{tag}/x": x, f"{tag}/t": t, f"{tag}/y": y, f"{tag}/gout": gout, f"{tag}/out": out,
                    f"{tag}/gx": gx, f"{tag}/grads": summarize(grads)})
    torch.save(rec, os.path.join(HERE, "dit_tiny.pt"))


def gen_unet_tiny():
    from models import unet as U
    from tools.nn import timestep_embedding
    rec = {}
    g = torch.Generator().manual_seed(4)
    rec["temb/t"] = torch.tensor([0.0, 1.0, 333.5, 999.0])
    rec["temb/out64"] = timestep_embedding(rec["temb/t"], 64)
    rec["temb/out33"] = timestep_embedding(rec["temb/t"], 33)
    # full tiny models: new attention order + legacy order, class-cond + uncond
    cfgs = {
        "new": dict(image_size=16, in_channels=3, model_channels=32, out_channels=3, num_res_blocks=1,
                    attention_resolutions=(2,), channel_mult=(1, 2), num_classes=10, num_heads=2,
                    use_scale_shift_norm=True, resblock_updown=True, use_new_attention_order=True),
        "legacy_ss": dict(image_size=16, in_channels=3, model_channels=32, out_channels=3, num_res_blocks=1,
                          attention_resolutions=(1, 2), channel_mult=(1, 2), num_classes=0, num_heads=1, num_head_channels=16,
                          use_scale_shift_norm=True, resblock_updown=True, use_new_attention_order=False),
        "legacy": dict(image_size=16, in_channels=3, model_channels=32, out_channels=6, num_res_blocks=1,
                       attention_resolutions=(1, 2), channel_mult=(1, 3), num_classes=0, num_heads=1,
                       num_head_channels=16, use_scale_shift_norm=False, resblock_updown=False,
                       use_new_attention_order=False),
    }
    for tag, kw in cfgs.items():
        torch.manual_seed(21)
        m = U.UNetModel(**kw)
        rec[f"{tag}/init_sd"] = summarize(m.state_dict())
        m.train()
        perturb_(m, 77, std=0.03)
        x = torch.randn(2, 3, 16, 16, generator=g)
        t = torch.tensor([12.0, 845.0])
        y = torch.tensor([3, 7]) if kw["num_classes"] else None
        gout = torch.randn(2, kw["out_channels"], 16, 16, generator=g)
        out, gx, grads = fwd_bwd(m, x, t, y, gout)
        rec.update({f"{tag}/kw": kw, f"{tag}/sd": summarize(m.state_dict()), f"{tag}/x": x,
                    f"{tag}/t": t, f"{tag}/y": y if y is not None else torch.zeros(0), f"{tag}/gout": gout,
                    f"{tag}/out": out, f"{tag}/gx": gx, f"{tag}/grads": summarize(grads)})
    # factory topology check: parameter counts of the presets used by BASELINE configs
    rec["nparams/UNet_64_uncond"] = sum(p.numel() for p in U.UNet_64(class_cond=False).parameters())
    rec["nparams/ADM_64_c1000"] = sum(p.numel() for p in U.ADM_64(num_classes=1000, class_cond=True).parameters())
    torch.save(rec, os.path.join(HERE, "unet_tiny.pt"))


def gen_unet_dropout():
    """nn.Dropout(0.1) inside the ResBlocks (--dropout, main.py:99; models/unet.py:206-213): training-mode forward / backward
    under torch.manual_seed(5) (the masks come from the CPU generator), plus the eval-mode forward."""
    from models import unet as U
    kw = dict(image_size=16, in_channels=3, model_channels=32, out_channels=3, num_res_blocks=1, attention_resolutions=(2,),
              channel_mult=(1, 2), num_classes=10, num_heads=2, dropout=0.1, use_scale_shift_norm=True, resblock_updown=True,
              use_new_attention_order=True)
    g = torch.Generator().manual_seed(14)
    torch.manual_seed(21)
    m = U.UNetModel(**kw)
    m.train()
    perturb_(m, 77, std=0.03)
    x = torch.randn(2, 3, 16, 16, generator=g)
    t, y = torch.tensor([12.0, 845.0]), torch.tensor([3, 7])
    gout = torch.randn(2, 3, 16, 16, generator=g)
    torch.manual_seed(5)
    out, gx, grads = fwd_bwd(m, x, t, y, gout)
    m.eval()
    with torch.no_grad():
        out_eval = m(x, t, y=y)
    torch.save({"kw": kw, "x": x, "t": t, "y": y, "gout": gout, "out": out, "gx": gx, "grads": summarize(grads),
                "out_eval": out_eval}, os.path.join(HERE, "unet_dropout.pt"))


def synth_loader(B, C, H, n_batches, num_classes, seed=123, latent=False):
    g = torch.Generator().manual_seed(seed)
    batches = []
    for _ in range(n_batches):
        if latent:
            x = torch.cat([torch.randn(B, C // 2, H, H, generator=g) * 4,
                           torch.rand(B, C // 2, H, H, generator=g) * 1.45 + 0.05], dim=1)
        else:
            x = torch.rand(B, C, H, H, generator=g) * 2 - 1
        y = torch.randint(0, max(num_classes, 1), (B,), generator=g)
        batches.append((x, y))
    return batches


def run_trainer(make_model, args, batches, steps, betas2=(0.9, 0.95), var_type="FIXED_LARGE"):
    import copy
    import random
    from tools import gaussian_diffusion as gd
    from tools.trainer import Trainer
    from tools.utils import get_lr_lambda
    random.seed(42); np.random.seed(42); torch.manual_seed(42)
    model = make_model()
    ema_model = copy.deepcopy(model)
    opt = torch.optim.AdamW(model.parameters(), lr=args.lr, betas=betas2, weight_decay=0.0, eps=1e-8)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=get_lr_lambda(args))
    diff = gd.GaussianDiffusion(args=args, betas=gd.get_named_beta_schedule(args.path_type, 1000),
                                model_mean_type=gd.ModelMeanType.EPSILON, model_var_type=gd.ModelVarType[var_type],
                                loss_type=gd.LossType.MSE, rescale_timesteps=True, device="cpu")
    tr = Trainer(args, torch.device("cpu"), model, ema_model, opt, sched, diff, batches, Pbar())
    losses = [tr.train_step(s) for s in range(1, steps + 1)]
    psum = float(sum(p.double().abs().sum() for p in model.parameters()))
    esum = float(sum(v.double().abs().sum() for v in ema_model.state_dict().values()))
    return {"losses": losses, "param_abs_sum": psum, "ema_abs_sum": esum, "lr_last": sched.get_last_lr()[0]}


def gen_trainer():
    from models import unet as U
    from models.dit import DiT
    rec = {}
    cfg1 = lambda: U.UNetModel(32, 3, 64, 3, 2, attention_resolutions=(), channel_mult=(1, 2, 2, 2), num_heads=4,
                               use_scale_shift_norm=True, resblock_updown=True, use_new_attention_order=True)
    b16 = synth_loader(16, 3, 32, 4, 0)
    rec["cfg1"] = run_trainer(cfg1, base_args(), b16, 5)
    rec["cfg1_accum2_clip"] = run_trainer(cfg1, base_args(grad_accumulation=2, grad_clip=1.0), b16, 3)
    rec["cfg1_warmup_cosine_minsnr"] = run_trainer(
        cfg1, base_args(weight_type="min_snr_5", warmup_steps=2, cosine_decay=True, total_steps=10, final_lr=1e-5),
        b16, 4)
    tiny_dit = lambda: DiT(image_size=8, patch_size=2, in_channels=4, hidden_size=64, depth=2, num_heads=2,
                           class_dropout_prob=0.0, num_classes=10, learn_sigma=False)
    lat = synth_loader(8, 8, 8, 3, 10, latent=True)
    rec["dit_tiny_latent"] = run_trainer(tiny_dit, base_args(in_chans=4, class_cond=True, dataset="Latent",
                                                              image_size=8, lr=1e-3), lat, 6)
    # DiT-B/4 (BASELINE config 4) at a reduced batch: seeded build, 3 steps
    dit_b4 = lambda: DiT(image_size=32, patch_size=4, in_channels=4, hidden_size=768, depth=12, num_heads=12,
                         class_dropout_prob=0.0, num_classes=1000, learn_sigma=False)
    latb = synth_loader(8, 8, 32, 3, 1000, latent=True)
    rec["dit_b4_b8"] = run_trainer(dit_b4, base_args(in_chans=4, class_cond=True, dataset="Latent", image_size=32),
                                   latb, 3)
    json.dump(rec, open(os.path.join(HERE, "trainer.json"), "w"), indent=1)


def gen_trainer_vb():
    """Trainer trajectories with the learned-variance objective (learn_sigma=True, LEARNED_RANGE: loss = mse + vb)."""
    from models import unet as U
    from models.dit import DiT
    rec = {}
    tiny_dit = lambda: DiT(image_size=8, patch_size=2, in_channels=4, hidden_size=64, depth=2, num_heads=2,
                           class_dropout_prob=0.0, num_classes=10, learn_sigma=True)
    lat = synth_loader(8, 8, 8, 3, 10, latent=True)
    rec["dit_tiny_learn_sigma"] = run_trainer(tiny_dit, base_args(in_chans=4, class_cond=True, dataset="Latent", image_size=8,
                                                                   lr=1e-3, learn_sigma=True), lat, 6, var_type="LEARNED_RANGE")
    tiny_unet = lambda: U.UNetModel(16, 3, 32, 6, 1, attention_resolutions=(2,), channel_mult=(1, 2), num_heads=2,
                                    use_scale_shift_norm=True, resblock_updown=True, use_new_attention_order=True)
    b8 = synth_loader(8, 3, 16, 3, 0)
    rec["unet_tiny_learn_sigma"] = run_trainer(tiny_unet, base_args(image_size=16, lr=1e-3, learn_sigma=True), b8, 5,
                                               var_type="LEARNED_RANGE")
    json.dump(rec, open(os.path.join(HERE, "trainer_vb.json"), "w"), indent=1)


BIG_CONFIGS = {
    # BASELINE.json configs 2, 3, 5 at batch 2 (the full batches are exercised on the GPU through size-independent properties)
    "unet64": dict(kind="unet", factory="UNet_64", kw=dict(class_cond=False), size=64, chans=3, classes=0),
    "adm64": dict(kind="unet", factory="ADM_64", kw=dict(num_classes=1000, class_cond=True), size=64, chans=3, classes=1000),
    "dit_xl2": dict(kind="dit", factory="DiT_XL", kw=dict(image_size=32, patch_size=2, in_channels=4, class_dropout_prob=0.0,
                                                           num_classes=1000, learn_sigma=False), size=32, chans=8, classes=1000),
}


def gen_bigcfg():
    """Full-size models of BASELINE configs 2, 3 and 5, rebuilt from a seed (torch.manual_seed(42) + the constructor, then
    perturb_(model, 7) so that zero-initialised layers carry gradient): per-sample terms['mse'], fingerprints of every
    parameter gradient of loss = mse.mean(), and two Trainer steps.  Reference: models/unet.py:993,1013, models/dit.py:373,
    tools/gaussian_diffusion.py:834-930, tools/trainer.py:68-150."""
    import random
    from models import unet as U
    from models import dit as D
    from tools import gaussian_diffusion as gd
    out = {}
    for name, c in BIG_CONFIGS.items():
        latent = c["kind"] == "dit"

        def make_model(c=c):
            m = getattr(U if c["kind"] == "unet" else D, c["factory"])(**c["kw"])
            perturb_(m, 7, std=0.02)
            return m
        args = base_args(in_chans=4 if latent else 3, class_cond=bool(c["classes"]), dataset="Latent" if latent else "ImageNet",
                         image_size=c["size"])
        random.seed(42); np.random.seed(42); torch.manual_seed(42)
        model = make_model()
        g = torch.Generator().manual_seed(31)
        Cx = 4 if latent else 3
        x = torch.randn(2, Cx, c["size"], c["size"], generator=g) * (0.7 if latent else 0.5)
        noise = torch.randn(2, Cx, c["size"], c["size"], generator=g)
        t = torch.tensor([37, 912])
        y = torch.tensor([3, 998]) if c["classes"] else None
        diff = gd.GaussianDiffusion(args=args, betas=gd.get_named_beta_schedule("cosine", 1000),
                                    model_mean_type=gd.ModelMeanType.EPSILON, model_var_type=gd.ModelVarType.FIXED_LARGE,
                                    loss_type=gd.LossType.MSE, rescale_timesteps=True, device="cpu")
        terms = diff.training_losses(model, x, None, t=t, model_kwargs={"y": y} if y is not None else {}, noise=noise)
        terms["loss"].mean().backward()
        rec = {"x": x, "noise": noise, "t": t, "y": y if y is not None else torch.zeros(0, dtype=torch.long),
               "mse": terms["mse"].detach().double(), "n_params": sum(p.numel() for p in model.parameters()),
               "params": summarize({k: v for k, v in model.named_parameters()}),
               "grads": summarize({k: v.grad for k, v in model.named_parameters() if v.grad is not None})}
        del model, terms
        loader = synth_loader(2, c["chans"], c["size"], 2, c["classes"], latent=latent)
        rec["trainer"] = run_trainer(make_model, args, loader, 2)
        out[name] = rec
        print("  bigcfg", name, rec["n_params"], [round(float(v), 6) for v in rec["mse"]], rec["trainer"]["losses"], flush=True)
    torch.save(out, os.path.join(HERE, "bigcfg.pt"))


def sampling_model(x, t, **kw):
    """Deterministic stand-in denoiser for the sampling goldens: bounded, depends on x, t (as the wrapped model sees it) and y;
    2C channels when asked (second half = variance values in [-1, 1])."""
    tt = t.float().view(-1, 1, 1, 1)
    m = 0.6 * torch.tanh(x) + 0.1 * torch.sin(tt * 0.01)
    if "y" in kw and kw["y"] is not None:
        m = m + 0.02 * kw["y"].view(-1, 1, 1, 1).float()
    return m


def sampling_model_2c(x, t, **kw):
    m = sampling_model(x, t, **kw)
    return torch.cat([m, 0.8 * torch.cos(3.0 * x)], dim=1)


def gen_sampling(gd):
    """Sampling side: space_timesteps, SpacedDiffusion wrapping, p_sample / ddim_sample loops with
    the CPU RNG stream, IntervalCFG.  Reference: tools/respace.py, gaussian_diffusion.py:278-384,461-560,603-790,
    tools/sampler.py:10-48."""
    from tools import respace as R
    from tools.sampler import IntervalCFG
    out = {}
    out["space"] = {str(k): sorted(R.space_timesteps(1000, k)) for k in ("ddim25", "ddim50", "10", "10,15,20", "250")}
    out["space_300_10_15_20"] = sorted(R.space_timesteps(300, [10, 15, 20]))
    shape = (3, 3, 8, 8)
    y = torch.tensor([1, 5, 9])
    res = {}
    for name, sched, mt, vt, respacing, kind, eta, clip in [
        ("ddim25_eps_fixed", "cosine", "EPSILON", "FIXED_LARGE", "ddim25", "ddim", 0.0, True),
        ("ddim10_eps_range_eta", "linear", "EPSILON", "LEARNED_RANGE", "10", "ddim", 0.7, True),
        ("ddim20_x0_small_noclip", "cosine", "START_X", "FIXED_SMALL", "20", "ddim", 0.0, False),
        ("p50_eps_range", "linear", "EPSILON", "LEARNED_RANGE", "50", "p", 0.0, True),
        ("p20_x0_large", "cosine", "START_X", "FIXED_LARGE", "20", "p", 0.0, True),
        ("p15_eps_learned", "cosine", "EPSILON", "LEARNED", "15", "p", 0.0, False),
    ]:
        learned = vt.startswith("LEARNED")
        d = R.SpacedDiffusion(use_timesteps=R.space_timesteps(1000, respacing), args=base_args(learn_sigma=learned, amp=False),
                              betas=gd.get_named_beta_schedule(sched, 1000), model_mean_type=gd.ModelMeanType[mt],
                              model_var_type=gd.ModelVarType[vt], loss_type=gd.LossType.MSE, rescale_timesteps=True, device="cpu")
        model = sampling_model_2c if learned else sampling_model
        torch.manual_seed(123)
        loop = d.ddim_sample_loop_progressive if kind == "ddim" else d.p_sample_loop_progressive
        kw = dict(clip_denoised=clip, model_kwargs={"y": y}, device="cpu")
        if kind == "ddim":
            kw["eta"] = eta
        traj = [o["sample"].clone() for o in loop(model, shape, **kw)]
        res[name] = {"final": traj[-1], "first": traj[0], "mid": traj[len(traj) // 2], "n": len(traj),
                     "timestep_map": torch.tensor(d.timestep_map)}
    out["loops"] = res
    # IntervalCFG: guidance on/off by interval, null label = num_classes
    x = torch.randn(4, 3, 8, 8, generator=torch.Generator().manual_seed(5))
    yy = torch.tensor([0, 3, 7, 9])
    cfg = {}
    for nm, scale, interval, tval in [("plain", 1.0, (-1.0, -1.0), 500.0), ("always", 2.5, (-1.0, -1.0), 500.0),
                                      ("inside", 1.8, (100.0, 600.0), 300.0), ("outside", 1.8, (100.0, 600.0), 800.0)]:
        m = IntervalCFG(sampling_model, 10, scale, interval, True)
        cfg[nm] = m(x, torch.full((4,), tval), y=yy)
    out["cfg"] = cfg
    out["cfg_x"], out["cfg_y"] = x, yy
    torch.save(out, os.path.join(HERE, "sampling.pt"))


def gen_samplers(gd):
    """EDM Euler / Heun sampler over the DDPM chain (tools/cfg_edm.py Net + ablation_sampler) and the FlowMatching SDE sampler
    (tools/gaussian_diffusion.py:1374-1409), driven with the deterministic stand-in denoiser and the CPU RNG stream."""
    from tools.cfg_edm import Net, ablation_sampler
    out = {"edm": {}, "flow_sde": {}}
    y = torch.tensor([1, 5, 9])
    lat = torch.randn(3, 3, 8, 8, generator=torch.Generator().manual_seed(8))
    for name, kw_net, kw_s in [
        ("edm_heun_cosine_eps", dict(pred_type="EPSILON", noise_schedule="cosine"), dict(num_steps=9, solver="heun")),
        ("edm_euler_linear_v", dict(pred_type="VELOCITY", noise_schedule="linear"), dict(num_steps=12, solver="euler")),
        ("vp_heun_logsnr_x0", dict(pred_type="START_X", noise_schedule="linear_logsnr"),
         dict(num_steps=8, solver="heun", discretization="vp", schedule="vp", scaling="vp")),
        ("iddpm_heun_churn", dict(pred_type="EPSILON", noise_schedule="cosine"),
         dict(num_steps=10, solver="heun", discretization="iddpm", S_churn=4.0, S_min=0.05, S_max=50.0, S_noise=1.003)),
        ("ve_euler", dict(pred_type="EPSILON", noise_schedule="linear"), dict(num_steps=7, solver="euler", discretization="ve", schedule="ve")),
    ]:
        net = Net(model=sampling_model, img_channels=3, img_resolution=8, label_dim=10, amp=False, **kw_net)
        torch.manual_seed(321)
        x = ablation_sampler(net, latents=lat, class_labels=y, **kw_s)
        out["edm"][name] = {"net": kw_net, "sampler": kw_s, "x": x, "sigma_min": net.sigma_min, "sigma_max": net.sigma_max,
                            "u_sample": net.u[::100].clone()}
    out["edm_latents"], out["y"] = lat, y
    noise = torch.randn(3, 3, 8, 8, generator=torch.Generator().manual_seed(9))
    for path in ("linear", "cosine"):
        for mt in ("EPSILON", "START_X", "VELOCITY", "VECTOR"):
            for solver in ("euler", "heun"):
                fm = gd.FlowMatching(args=base_args(path_type=path, sampler_type="sde"), model_mean_type=gd.ModelMeanType[mt], device="cpu")
                torch.manual_seed(77)
                out["flow_sde"][f"{path}/{mt}/{solver}"] = fm.sde_sample(sampling_model, noise, "cpu", num_steps=9, solver=solver, y=y)
    out["flow_noise"] = noise
    torch.save(out, os.path.join(HERE, "samplers.pt"))


def gen_misc():
    from tools.utils import warmup_cosine_lr
    from tools import resample as R
    from tools.trainer import sample_from_latent, ema
    rec = {}
    rec["lr"] = [[s, warmup_cosine_lr(s, 5, 50, 1e-4, 1e-6, True), warmup_cosine_lr(s, 5, 50, 1e-4, 1e-6, False),
                  warmup_cosine_lr(s, 0, 50, 1e-4, 0.0, True)] for s in range(0, 51, 3)]
    diff = SimpleNamespace(num_timesteps=20)
    s = R.create_named_schedule_sampler("loss-second-moment", diff)
    rng = np.random.RandomState(0)
    w_before = s.weights().tolist()
    for _ in range(15):
        ts = list(range(20))
        s.update_with_all_losses(ts, (rng.rand(20) * (1 + np.arange(20))).tolist())
    rec["lsm_weights_before"] = w_before
    rec["lsm_weights_after"] = s.weights().tolist()
    np.random.seed(3)
    idx, w = s.sample(16, "cpu")
    rec["lsm_sample_idx"] = idx.tolist(); rec["lsm_sample_w"] = tolist(w)
    u = R.create_named_schedule_sampler("uniform", diff)
    np.random.seed(3)
    idx, w = u.sample(8, "cpu")
    rec["uni_sample_idx"] = idx.tolist(); rec["uni_sample_w"] = tolist(w)
    g = torch.Generator().manual_seed(9)
    lat = torch.randn(2, 8, 4, 4, generator=g)
    torch.manual_seed(1)
    rec["sfl_in"] = tolist(lat); rec["sfl_out"] = tolist(sample_from_latent(lat, 0.18215))
    json.dump(rec, open(os.path.join(HERE, "misc.json"), "w"))


def gen_uvit_anchor():
    """The reference's OWN statement of ViT attention and MLP (models/uvit.py:55-93 `Attention` in its 'math' and 'flash' modes,
    tools/timm.py:96-112 `Mlp`): seeded weights, input, output and input gradient.  tests/test_oracle_goldens.py loads the same
    weights into oracle/timm_restatement.{Attention, Mlp} -- the stand-in for the absent timm==0.9.2 classes of models/dit.py:17 --
    and must reproduce these numbers: a reference-held pin of the (K H D) qkv packing and the hd^-1/2 scale."""
    import models.uvit as uvit
    from tools.timm import Mlp as RefMlp
    out = {}
    for heads, dim, L, B in ((4, 64, 16, 2), (6, 96, 9, 3)):
        torch.manual_seed(100 + heads)
        att = uvit.Attention(dim, num_heads=heads, qkv_bias=True)
        perturb_(att, 7 + heads)
        mlp = RefMlp(dim, 4 * dim, act_layer=lambda: torch.nn.GELU(approximate="tanh"))
        perturb_(mlp, 17 + heads)
        x = torch.randn(B, L, dim, generator=torch.Generator().manual_seed(3 + heads))
        gy = torch.randn(B, L, dim, generator=torch.Generator().manual_seed(5 + heads))
        rec = {"x": x, "gy": gy, "attn_state": {k: v.clone() for k, v in att.state_dict().items()},
               "mlp_state": {k: v.clone() for k, v in mlp.state_dict().items()}}
        for mode in ("math", "flash"):
            uvit.ATTENTION_MODE = mode
            xi = x.clone().requires_grad_(True)
            y = att(xi)
            (gx,) = torch.autograd.grad(y, xi, gy)
            rec[f"attn_{mode}_y"], rec[f"attn_{mode}_gx"] = y.detach(), gx
        xi = x.clone().requires_grad_(True)
        y = mlp(xi)
        (gx,) = torch.autograd.grad(y, xi, gy)
        rec["mlp_y"], rec["mlp_gx"] = y.detach(), gx
        out[f"h{heads}_d{dim}"] = rec
    torch.save(out, os.path.join(HERE, "uvit_anchor.pt"))


def main():
    install_stubs()
    torch.set_num_threads(8)
    from tools import gaussian_diffusion as gd
    jobs = {"tables": lambda: gen_tables(gd), "weights": lambda: gen_loss_weights(gd),
            "objective": lambda: gen_objective(gd), "dit": gen_dit_tiny, "unet": gen_unet_tiny, "unet_dropout": gen_unet_dropout, "misc": gen_misc,
            "trainer": gen_trainer, "bigcfg": gen_bigcfg, "vb": lambda: gen_vb(gd), "trainer_vb": gen_trainer_vb, "sampling": lambda: gen_sampling(gd), "samplers": lambda: gen_samplers(gd), "uvit_anchor": gen_uvit_anchor}
    for name in (sys.argv[1:] or list(jobs)):
        jobs[name]()
        print("wrote", name)


if __name__ == "__main__":
    main()
