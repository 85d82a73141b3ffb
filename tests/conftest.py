import json
import os
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(REPO, "tests", "golden")
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def base_args(**kw):
    a = dict(weight_type="lambda", gamma=0.0, learn_sigma=False, p2_gamma=1, p2_k=1, time_dist=["uniform", -0.8, 0.8],
             learn_align=False, align_type="mse", amp=False, dataset="CIFAR-10", class_cond=False, parallel=False,
             grad_accumulation=1, in_chans=3, latent_scale=0.18215, grad_clip=None, ema_decay=0.9999,
             enc_type="dinov2-vit-b", image_size=32, path_type="cosine", sampler_type="ode", lr=1e-4, final_lr=0.0,
             warmup_steps=0, total_steps=1000, cosine_decay=False)
    a.update(kw)
    return SimpleNamespace(**a)


def load_pt(name):
    return torch.load(os.path.join(GOLDEN, name), weights_only=True)


def load_json(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def perturb_(model, seed, std=0.05):
    """Same deterministic perturbation as tests/golden/make_goldens.py."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in model.parameters():
            if p.requires_grad:
                p.add_((torch.randn(p.shape, generator=g) * std).to(p.device))


def fingerprint(v):
    f = v.detach().double().flatten().cpu()
    stride = max(1, f.numel() // 64)
    return torch.stack([f.sum(), f.abs().sum(), f.norm()]), f[::stride][:64].clone()


def assert_fingerprints(tensors, golden, rtol, atol, what=""):
    assert set(tensors) == set(golden), f"{what}: key mismatch {set(tensors) ^ set(golden)}"
    for k, v in tensors.items():
        stats, sample = fingerprint(v)
        g = golden[k]
        scale = float(g["stats"][2]) / max(1.0, float(v.numel()) ** 0.5)  # rms of the golden tensor
        torch.testing.assert_close(sample, g["sample"], rtol=rtol, atol=atol + rtol * scale, msg=lambda m: f"{what}:{k} sample {m}")
        torch.testing.assert_close(stats[2], g["stats"][2], rtol=rtol * 10, atol=atol, msg=lambda m: f"{what}:{k} l2 {m}")


def synth_loader(B, C, H, n_batches, num_classes, seed=123, latent=False):
    """SURVEY §8(d) synthetic batches; identical to make_goldens.synth_loader."""
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


class Pbar:
    def update(self, n):
        pass

    def set_postfix(self, **kw):
        pass


def fake_model(x, t, **kw):
    return 0.5 * x + 1e-3 * t.view(-1, 1, 1, 1).float() + (0.01 * kw["y"].view(-1, 1, 1, 1).float() if "y" in kw else 0)


def sampling_model(x, t, **kw):
    """The stand-in denoiser of tests/golden/make_goldens.py::gen_sampling (same arithmetic, any device)."""
    tt = t.float().view(-1, 1, 1, 1)
    m = 0.6 * torch.tanh(x) + 0.1 * torch.sin(tt * 0.01)
    if "y" in kw and kw["y"] is not None:
        m = m + 0.02 * kw["y"].view(-1, 1, 1, 1).float()
    return m


def sampling_model_2c(x, t, **kw):
    return torch.cat([sampling_model(x, t, **kw), 0.8 * torch.cos(3.0 * x)], dim=1)


SAMPLING_CASES = [  # name, schedule, mean type, var type, respacing, kind, eta, clip_denoised
    ("ddim25_eps_fixed", "cosine", "EPSILON", "FIXED_LARGE", "ddim25", "ddim", 0.0, True),
    ("ddim10_eps_range_eta", "linear", "EPSILON", "LEARNED_RANGE", "10", "ddim", 0.7, True),
    ("ddim20_x0_small_noclip", "cosine", "START_X", "FIXED_SMALL", "20", "ddim", 0.0, False),
    ("p50_eps_range", "linear", "EPSILON", "LEARNED_RANGE", "50", "p", 0.0, True),
    ("p20_x0_large", "cosine", "START_X", "FIXED_LARGE", "20", "p", 0.0, True),
    ("p15_eps_learned", "cosine", "EPSILON", "LEARNED", "15", "p", 0.0, False),
]
