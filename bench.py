#!/usr/bin/env python3
"""Headline benchmark: training images/sec of the variance-aware-weighted diffusion step on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload dit_b4]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

One "step" = one full `Trainer.train_step` on one synthetic batch already resident in HBM:
latent sampling, noise + timestep draw, q_sample, DiT forward, weighted-MSE loss, hand-written backward,
(N>1: bucketed RCCL all-reduce overlapped with backward), fused AdamW + EMA.  Nothing is skipped or cached.
Default workload = BASELINE.json config 4, the one its MFMA target is quoted on: DiT-B/4 on 4x32x32 latents,
batch 256 PER GPU (weak scaling: global batch = 256*N), bf16 MFMA with f32 accumulation, weight_type 'lambda'.

Prints ONE JSON line on rank 0 (contract in the project brief) with two extra objects:
  roofline      the dominant kernel (bf16 MFMA GEMM): algorithmic FLOP / HIP-event launch time, vs 2.5 PFLOP/s
  cpu_baseline  the CPU oracle (oracle/, kind "port") timed on this host on a bounded sample of the same workload
"""
import argparse
import copy
import json
import os
import sys
import time
from types import SimpleNamespace

import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

WORKLOADS = {
    # per-GPU batch and train GFLOP/img from BASELINE.md §3 (BASELINE.json configs 4, 1, 2, 3)
    "dit_b4": dict(kind="dit", model="DiT-B", patch=4, batch=256, gflop_per_img=33.37, desc="DiT-B/4, 4x32x32 latents, 1000 classes"),
    "dit_b2": dict(kind="dit", model="DiT-B", patch=2, batch=256, gflop_per_img=138.0, desc="DiT-B/2, 4x32x32 latents, 1000 classes"),
    "dit_xl2": dict(kind="dit", model="DiT-XL", patch=2, batch=128, gflop_per_img=711.7,
                    desc="DiT-XL/2, 4x32x32 latents, 1000 classes (BASELINE config 5 at its per-GPU batch 1024/8; bf16, not fp8)"),
    "dit_s4": dict(kind="dit", model="DiT-S", patch=4, batch=256, gflop_per_img=None, desc="DiT-S/4 (smoke)"),
    "unet32": dict(kind="unet", size=32, classes=0, batch=16, gflop_per_img=9.91,
                   desc="CIFAR-10-shaped UNet (32x32, base 64 ch, mult 1,2,2,2, 10.4 M params), BASELINE config 1"),
    "unet64": dict(kind="unet", size=64, classes=0, batch=128, gflop_per_img=464.7,
                   desc="UNet_64 (CelebA-64, 192 ch, mult 1,2,2,2, attention at 16/8, 128 M params), BASELINE config 2"),
    "adm64": dict(kind="unet", size=64, classes=1000, batch=256, gflop_per_img=657.9,
                  desc="ADM_64 (ImageNet-64, 192 ch, mult 1,2,3,4, attention at 32/16/8, 296 M params), BASELINE config 3"),
}
BF16_MFMA_PEAK_TFLOPS = 2500.0   # dense, /opt/skills/guides/MI355X_MICROARCH.md


def pmc_traffic(workload, batch):
    """HBM bytes per GEMM launch from the PMC counters (FETCH_SIZE x 2 + WRITE_SIZE, KiB -> bytes; the gfx950 corrections of
    MI355X_MICROARCH.md).  Counters cannot be read from inside this process: they come from two separate rocprofv3 --pmc
    passes of this same command, folded by tools/pmc_traffic.py into profiles/ (named per round); null when no such file
    exists for the workload / batch being run."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", f"r01_final_{workload}_bs{batch}_bf16_hbm_traffic.json")
    try:
        with open(path) as f:
            return round(json.load(f)["hbm_bytes_per_launch"])
    except (OSError, KeyError, ValueError):
        return None


def make_args(**kw):
    a = dict(weight_type="lambda", gamma=0.0, learn_sigma=False, p2_gamma=1, p2_k=1, time_dist=["uniform", -0.8, 0.8],
             learn_align=False, align_type="mse", amp=True, dataset="Latent", class_cond=True, parallel=False,
             grad_accumulation=1, in_chans=4, latent_scale=0.18215, grad_clip=None, ema_decay=0.9999,
             image_size=32, path_type="cosine", lr=1e-4, final_lr=0.0, warmup_steps=0, total_steps=400000,
             cosine_decay=False, defer_loss_sync=True)
    a.update(kw)
    return SimpleNamespace(**a)


def workload_args(wl, **kw):
    if wl["kind"] == "unet":
        return make_args(in_chans=3, dataset="CelebA", image_size=wl["size"], class_cond=bool(wl["classes"]), **kw)
    return make_args(**kw)


def synth_batches(B, n, device, seed, wl=None):
    """latents = cat[mean ~ 4*N(0,1), std ~ U(0.05,1.5)] [B,8,32,32], labels in [0,1000);
    pixel models: x = rand*2-1 in [B,3,H,W]."""
    g = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(n):
        if wl is not None and wl["kind"] == "unet":
            x = torch.rand(B, 3, wl["size"], wl["size"], generator=g) * 2 - 1
        else:
            x = torch.cat([torch.randn(B, 4, 32, 32, generator=g) * 4, torch.rand(B, 4, 32, 32, generator=g) * 1.45 + 0.05], 1)
        y = torch.randint(0, 1000, (B,), generator=g)
        out.append((x.to(device), y.to(device)))
    return out


class _Sampler:
    def set_epoch(self, e):
        pass


class _Loader(list):
    sampler = _Sampler()


def make_model(pkg, wl):
    """pkg is vaw_amd (HIP) or a namespace with the oracle's constructors (CPU baseline)."""
    if wl["kind"] == "dit":
        return pkg.DiT_models[wl["model"]](image_size=32, patch_size=wl["patch"], in_channels=4, class_dropout_prob=0.0,
                                           num_classes=1000, learn_sigma=False)
    if wl["size"] == 32:
        return pkg.UNetModel(32, 3, 64, 3, 2, attention_resolutions=(), channel_mult=(1, 2, 2, 2), num_heads=4,
                             use_scale_shift_norm=True, resblock_updown=True, use_new_attention_order=True)
    if wl["classes"]:
        return pkg.ADM_64(num_classes=1000, class_cond=True)
    return pkg.UNet_64(class_cond=False)


def build(pkg, wl, args, device, rank):
    torch.manual_seed(42)          # same seed on every rank => identical replicas, like DDP's broadcast
    model = make_model(pkg, wl).to(device)
    # random-init weights of the architecture; adaLN-Zero would make every block an identity, so perturb
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for p in model.parameters():
            if p.requires_grad:
                p.add_((torch.randn(p.shape, generator=g) * 0.02).to(p.device))
    ema_model = copy.deepcopy(model)
    return model, ema_model


def cpu_baseline(wl, budget_s=25.0):
    """The CPU oracle (torch restatement of the reference path, pinned by tests/golden) on this host."""
    from oracle import diffusion as od, dit as odit, trainer as otr, unet as ounet
    torch.manual_seed(42)
    # a one-GPU box gives this job a 16-core CPU share; more threads than that only thrash
    torch.set_num_threads(int(os.environ.get("VAW_CPU_THREADS", "16")))
    B = {"dit": 32, "unet": 16 if wl.get("size") == 32 else 4}[wl["kind"]]
    args = workload_args(wl, amp=False, defer_loss_sync=False)
    opkg = SimpleNamespace(DiT_models=odit.DiT_models, UNetModel=ounet.UNetModel, ADM_64=ounet.ADM_64, UNet_64=ounet.UNet_64)
    model = make_model(opkg, wl)
    ema_model = copy.deepcopy(model)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, betas=(0.9, 0.95), weight_decay=0.0)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=otr.get_lr_lambda(args))
    diff = od.GaussianDiffusion(args=args, betas=od.get_named_beta_schedule("cosine", 1000),
                                model_mean_type=od.ModelMeanType.EPSILON, model_var_type=od.ModelVarType.FIXED_LARGE,
                                loss_type=od.LossType.MSE, rescale_timesteps=True)
    tr = otr.Trainer(args, torch.device("cpu"), model, ema_model, opt, sched, diff, synth_batches(B, 2, "cpu", 123, wl))
    tr.train_step(0)               # warm-up (allocator, oneDNN primitive cache)
    t0, n = time.perf_counter(), 0
    while n < 2 or (time.perf_counter() - t0 < budget_s and n < 20):
        tr.train_step(n + 1)
        n += 1
    dt = time.perf_counter() - t0
    return {"value": round(B * n / dt, 2), "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} steps of {wl['desc']} at batch {B}, f32, oracle/ Trainer (fwd+bwd+AdamW+EMA), after 1 warm-up step"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="dit_b4", choices=list(WORKLOADS))
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (default: the workload's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-trace", action="store_true", help="do not bracket GEMM launches with HIP events")
    ap.add_argument("--fp32", action="store_true", help="parity-mode kernels (not the headline number)")
    ap.add_argument("--graph", action="store_true", help="capture the step into one hipGraph (Trainer args.hip_graph); implies --no-trace")
    a = ap.parse_args()

    import vaw_amd
    vaw_amd.lib()                   # fail loudly if the HIP library is missing
    wl = WORKLOADS[a.workload]
    B = a.batch or wl["batch"]
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}")
    # VAW_REHEARSE_ONE_GPU=1: every rank on cuda:0 over gloo -- exercises the N-rank control flow (buckets, side stream,
    # barriers, max-over-ranks timing) on a one-GPU box; the numbers it prints are NOT a multi-GPU measurement
    rehearse = os.environ.get("VAW_REHEARSE_ONE_GPU") == "1"
    if rehearse:
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    parallel = world > 1
    if parallel:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        vaw_amd.dist_util.setup_dist(backend="gloo" if rehearse else None, device_index=local)
    if a.graph:
        a.no_trace = True
    args = workload_args(wl, parallel=parallel, amp=not a.fp32, hip_graph=a.graph)
    model, ema_model = build(vaw_amd, wl, args, device, rank)
    if a.fp32:
        model.set_compute_dtype("fp32")
    net = vaw_amd.DistributedDataParallel(model) if parallel else model
    opt = vaw_amd.FusedAdamW(model, lr=args.lr, betas=(0.9, 0.95), weight_decay=0.0, eps=1e-8)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=vaw_amd.get_lr_lambda(args))
    diff = vaw_amd.GaussianDiffusion(args=args, betas=vaw_amd.get_named_beta_schedule("cosine", 1000),
                                     model_mean_type=vaw_amd.ModelMeanType.EPSILON,
                                     model_var_type=vaw_amd.ModelVarType.FIXED_LARGE, loss_type=vaw_amd.LossType.MSE,
                                     rescale_timesteps=True)
    torch.manual_seed(1000 + rank)          # seed + rank: every rank draws its own t / noise (reference utils.py:62-69)
    loader = _Loader(synth_batches(B, 4, device, 123 + rank, wl))
    tr = vaw_amd.Trainer(args, device, net, ema_model if rank == 0 else None, opt, sched, diff, loader)

    def barrier():
        if parallel:
            dist.barrier()
        torch.cuda.synchronize()

    losses = []
    for s in range(a.warmup):
        losses.append(tr.train_step(s))
    trace = None
    if not a.no_trace and rank == 0:
        trace = vaw_amd.ops.GemmTrace()
    # the HIP-event brackets around every GEMM call cost ~3 % of a step (they fence the launch stream), so inside the timed
    # region only every 4th step carries them (all steps when fewer than 8 are timed); roofline = those steps' launches
    stride = 4 if a.steps >= 8 else 1
    barrier()
    t0 = time.perf_counter()
    for s in range(a.steps):
        vaw_amd.ops.gemm_trace = trace if s % stride == 0 else None
        losses.append(tr.train_step(a.warmup + s))
    barrier()
    elapsed = time.perf_counter() - t0
    vaw_amd.ops.gemm_trace = None
    if parallel:
        tmax = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    last_loss = float(losses[-1])
    if not (last_loss == last_loss) or abs(last_loss) > 1e4:
        raise SystemExit(f"non-finite / diverged loss {last_loss}: the measurement is void")

    if rank == 0:
        ms = 1e3 * elapsed / a.steps
        ips = B * world * a.steps / elapsed
        rec = {"metric": "training images/sec", "value": round(ips, 2), "unit": "images/sec", "n_gpus": world,
               "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "f32" if a.fp32 else "bf16", "data": "synthetic",
               "config": {"workload": f"{wl['desc']}; Trainer.train_step: q_sample + fwd + lambda-weighted MSE + bwd + "
                                      f"AdamW + EMA; per-GPU batch {B}", "global_batch": B * world,
                          "parallelism": f"dp{world}", "weight_type": args.weight_type, "last_loss": round(last_loss, 5)}}
        if wl["gflop_per_img"]:
            rec["config"]["step_mfma_util_vs_2.5PF"] = round(ips / world * wl["gflop_per_img"] / 1e3 / BF16_MFMA_PEAK_TFLOPS, 4)
        if trace is not None:
            traced_steps = len(range(0, a.steps, stride))
            summ = trace.summarize()
            fast = {k: v for k, v in summ.items() if k.startswith("bf16_mfma")}
            if fast:
                flop = sum(v["flop"] for v in fast.values())
                t_ms = sum(v["ms"] for v in fast.values())
                n_l = sum(v["launches"] for v in fast.values())
                ach = flop / (t_ms * 1e-3) / 1e12
                rec["roofline"] = {
                    "kernel": "gemm_bf16_kernel (v_mfma_f32_16x16x32_bf16, 128x128x64 tiles; fwd/dgrad/wgrad variants)",
                    "bound": "mfma", "achieved": round(ach, 1), "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "traffic_unit": "HBM bytes per launch (PMC, separate passes)",
                    "frac": round(ach / BF16_MFMA_PEAK_TFLOPS, 4), "traffic": None if a.fp32 else pmc_traffic(a.workload, B),
                    "launches_per_step": n_l / traced_steps, "avg_launch_us": round(1e3 * t_ms / n_l, 2),
                    "traced_steps": traced_steps,
                    "gemm_share_of_step": round(t_ms / traced_steps / (1e3 * elapsed / a.steps), 4),
                    "by_variant": {k: {"launches_per_step": v["launches"] / traced_steps,
                                       "tflops": round(v["flop"] / (v["ms"] * 1e-3) / 1e12, 1),
                                       "avg_us": round(1e3 * v["ms"] / v["launches"], 2)} for k, v in summ.items()}}
        if world == 1 and not a.no_cpu_baseline:
            del tr, opt, model, ema_model, net
            torch.cuda.empty_cache()
            rec["cpu_baseline"] = cpu_baseline(wl)
        print(json.dumps(rec), flush=True)
    if parallel:
        dist.barrier()
        vaw_amd.dist_util.cleanup_dist()


if __name__ == "__main__":
    main()
