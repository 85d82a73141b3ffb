#!/usr/bin/env python3
"""Headline benchmark: training images/sec of the variance-aware-weighted diffusion step on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload dit_b4] [--scaling strong|weak]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W
(a plain `python bench.py --gpus N` starts its N ranks itself, before anything touches the GPU, and relays rank 0's line)

One "step" = one full `Trainer.train_step` on one synthetic batch already resident in HBM:
latent sampling, noise + timestep draw, q_sample, DiT forward, weighted-MSE loss, hand-written backward,
(N>1: bucketed RCCL all-reduce overlapped with backward), fused AdamW + EMA.  Nothing is skipped or cached.
Default workload = BASELINE.json config 4, the one its MFMA target is quoted on: DiT-B/4 on 4x32x32 latents, GLOBAL batch
256 sharded over the N GPUs as the reference shards it (main.py:166-180: per-GPU batch = batch_size // world_size; strong
scaling), bf16 MFMA with f32 accumulation, weight_type 'lambda'.  For N > 1 the same process then also times the weak
configuration (256 per GPU) and reports it in the extra field "weak".

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
    "dit_xl2_fp8": dict(kind="dit", model="DiT-XL", patch=2, batch=128, gflop_per_img=711.7, fp8=True,
                        desc="DiT-XL/2, 4x32x32 latents, 1000 classes (BASELINE config 5 at its per-GPU batch 1024/8): the blocks' Linear "
                             "layers on the scaled fp8 MFMA (e4m3 weights/activations, e5m2 gradients, per-tensor scales, f32 accumulate), "
                             "attention / norms / embedders in bf16"),
    "dit_s4": dict(kind="dit", model="DiT-S", patch=4, batch=256, gflop_per_img=None, desc="DiT-S/4 (smoke)"),
    "unet32": dict(kind="unet", size=32, classes=0, batch=16, gflop_per_img=9.91,
                   desc="CIFAR-10-shaped UNet (32x32, base 64 ch, mult 1,2,2,2, 10.4 M params), BASELINE config 1"),
    "unet64": dict(kind="unet", size=64, classes=0, batch=128, gflop_per_img=464.7,
                   desc="UNet_64 (CelebA-64, 192 ch, mult 1,2,2,2, attention at 16/8, 128 M params), BASELINE config 2"),
    "adm64": dict(kind="unet", size=64, classes=1000, batch=256, gflop_per_img=657.9,
                  desc="ADM_64 (ImageNet-64, 192 ch, mult 1,2,3,4, attention at 32/16/8, 296 M params), BASELINE config 3"),
}
BF16_MFMA_PEAK_TFLOPS = 2500.0   # dense, /opt/skills/guides/MI355X_MICROARCH.md
FP8_MFMA_PEAK_TFLOPS = 5000.0    # dense fp8 (v_mfma_scale_f32_16x16x128_f8f6f4), same guide


def pmc_traffic(workload, batch):
    """HBM bytes per GEMM launch from the PMC counters (FETCH_SIZE x 2 + WRITE_SIZE, KiB -> bytes; the gfx950 corrections of
    MI355X_MICROARCH.md).  Counters cannot be read from inside this process: they come from two separate rocprofv3 --pmc
    passes of this same command, folded by tools/pmc_traffic.py into profiles/r<NN>_*_hbm_traffic.json; the newest round's
    file for the workload / batch being run is used and named in the line ("traffic_source"); (None, None) without one."""
    import glob
    import re
    pat = os.path.join(REPO, "profiles", f"r*_{workload}_bs{batch}_*hbm_traffic.json")
    best = None
    for path in glob.glob(pat):
        m = re.match(r"r(\d+)_", os.path.basename(path))
        if m and (best is None or int(m.group(1)) >= best[0]):
            best = (int(m.group(1)), path)
    if best is None:
        return None, None
    try:
        with open(best[1]) as f:
            return round(json.load(f)["hbm_bytes_per_launch"]), os.path.relpath(best[1], REPO)
    except (OSError, KeyError, ValueError):
        return None, None


def pmc_collected(src):
    """Collection date recorded inside a committed traffic fold (tools/pmc_traffic.py writes it), else None."""
    if not src:
        return None
    try:
        with open(os.path.join(REPO, src)) as f:
            return json.load(f).get("collected")
    except (OSError, ValueError):
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


def usable_cores():
    """Cores this process may really use: the affinity mask, cut by the cgroup CPU quota (a container sees every host core in its
    mask but is throttled to its share: oversubscribing it with one thread per visible core stalls for minutes)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]                      # cgroup v2
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())                  # cgroup v1
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(wl, batch, budget_s=25.0):
    """The CPU oracle (torch restatement of the reference path, pinned by tests/golden) on this host: the workload's batch
    when one step fits the budget, else the largest batch that does (flagged in "reduced_batch")."""
    from oracle import diffusion as od, dit as odit, trainer as otr, unet as ounet
    torch.manual_seed(42)
    usable = usable_cores()
    # default: at most 16 threads, the CPU share of one GPU on the measurement boxes (VAW_CPU_THREADS overrides)
    threads = int(os.environ.get("VAW_CPU_THREADS", min(usable, 16)))
    torch.set_num_threads(threads)
    print(f"[bench] cpu_baseline: {threads} threads ({usable} usable, {os.cpu_count()} on the host)", file=sys.stderr, flush=True)
    args = workload_args(wl, amp=False, defer_loss_sync=False)
    opkg = SimpleNamespace(DiT_models=odit.DiT_models, UNetModel=ounet.UNetModel, ADM_64=ounet.ADM_64, UNet_64=ounet.UNet_64)
    model = make_model(opkg, wl)
    ema_model = copy.deepcopy(model)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, betas=(0.9, 0.95), weight_decay=0.0)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=otr.get_lr_lambda(args))
    diff = od.GaussianDiffusion(args=args, betas=od.get_named_beta_schedule("cosine", 1000),
                                model_mean_type=od.ModelMeanType.EPSILON, model_var_type=od.ModelVarType.FIXED_LARGE,
                                loss_type=od.LossType.MSE, rescale_timesteps=True)
    # probe one small step to size the sample: the timed steps run the config batch if ~3 of them fit the budget
    probe_b = min(batch, 8)
    tr = otr.Trainer(args, torch.device("cpu"), model, ema_model, opt, sched, diff, synth_batches(probe_b, 1, "cpu", 5, wl))
    tr.train_step(0)               # warm-up (allocator, oneDNN primitive cache)
    t0 = time.perf_counter()
    tr.train_step(1)
    per_img = (time.perf_counter() - t0) / probe_b
    B = batch
    while B > 1 and 3 * B * per_img > budget_s:
        B //= 2
    tr = otr.Trainer(args, torch.device("cpu"), model, ema_model, opt, sched, diff, synth_batches(B, 2, "cpu", 123, wl))
    t0, n = time.perf_counter(), 0
    while n < 2 or (time.perf_counter() - t0 < budget_s and n < 20):
        tr.train_step(n + 2)
        n += 1
        print(f"[bench] cpu_baseline: step {n} at batch {B}, {time.perf_counter() - t0:.1f} s", file=sys.stderr, flush=True)
    dt = time.perf_counter() - t0
    return {"value": round(B * n / dt, 2), "unit": "images/sec", "cores": threads, "host_cpu_count": os.cpu_count(),
            "usable_cores": usable, "kind": "port", "reduced_batch": None if B == batch else B,
            "sample": f"{n} steps of {wl['desc']} at batch {B}, f32, oracle/ Trainer (fwd+bwd+AdamW+EMA), {threads} threads, "
                      f"after 2 warm-up steps at batch {probe_b}"}


def visible_gpus():
    """GPU count WITHOUT any HIP call (torch.cuda.device_count() may reach hipGetDeviceCount, i.e. initialise the runtime in a
    parent that is about to start child processes): KFD topology nodes with SIMDs are GPUs; HIP_/ROCR_VISIBLE_DEVICES narrow them."""
    import glob
    n = 0
    for path in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            props = dict(line.split()[:2] for line in open(path) if len(line.split()) >= 2)
        except OSError:
            continue
        n += int(props.get("simd_count", "0")) > 0
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None and v.strip() != "":
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as child processes (this parent never touches the
    GPU or the HIP runtime), relay rank 0's JSON line, exit with the worst return code.  All children are polled: when one
    dies the others (parked in a collective) are terminated instead of waiting for the backend watchdog, and every rank's
    stderr is kept."""
    import socket
    import subprocess
    import tempfile
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    base = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(n), HSA_ENABLE_IPC_MODE_LEGACY="0")
    have = visible_gpus()
    if have < n and "VAW_REHEARSE_ONE_GPU" not in base:
        raise SystemExit(f"--gpus {n} but only {have} visible (VAW_REHEARSE_ONE_GPU=1 rehearses the "
                         f"{n}-rank control flow on one GPU over gloo; its numbers are not a multi-GPU measurement)")
    procs, logs = [], []
    out0 = tempfile.TemporaryFile(mode="w+")
    for r in range(n):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        err = tempfile.TemporaryFile(mode="w+")
        logs.append(err)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL, stderr=err, text=True))
    rc, live = 0, set(range(n))
    while live:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                rc = code
                print(f"[bench] rank {r} exited with {code}: stopping the other ranks", file=sys.stderr, flush=True)
                for o in live:
                    procs[o].terminate()
                deadline = time.time() + 10
                while any(procs[o].poll() is None for o in live) and time.time() < deadline:
                    time.sleep(0.2)
                for o in live:
                    if procs[o].poll() is None:
                        procs[o].kill()
        time.sleep(0.05)
    for r, err in enumerate(logs):
        err.seek(0)
        text = err.read()
        if text and (r == 0 or rc != 0):
            sys.stderr.write("".join(f"[rank {r}] {ln}\n" for ln in text.splitlines()) if r else text)
    out0.seek(0)
    # rank 0's stdout may carry backend chatter (e.g. gloo's connection notice): relay the result line alone on stdout
    for line in out0.read().splitlines():
        (sys.stdout if line.startswith('{"metric"') else sys.stderr).write(line + "\n")
    sys.stdout.flush()
    raise SystemExit(rc)


def timed_steps(tr, first_step, n_steps, barrier, parallel, device):
    """EXACTLY n_steps steps between barrier + synchronize on both sides; -> (seconds, MAX over ranks; per-step ms list of this
    rank from stream markers, which fence nothing)."""
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(n_steps + 1)]
    losses = []
    barrier()
    t0 = time.perf_counter()
    marks[0].record()
    for s in range(n_steps):
        losses.append(tr.train_step(first_step + s))
        marks[s + 1].record()
    barrier()
    elapsed = time.perf_counter() - t0
    if parallel:
        tmax = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    per = [marks[i].elapsed_time(marks[i + 1]) for i in range(n_steps)]
    return elapsed, per, losses


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="dit_b4", choices=list(WORKLOADS))
    ap.add_argument("--batch", type=int, default=None, help="GLOBAL batch of the strong configuration = per-GPU batch of the weak one (default: the workload's)")
    ap.add_argument("--scaling", default=None, choices=["strong", "weak"],
                    help="strong (default): global batch fixed, batch // N per GPU (reference main.py:166-180), plus a weak pass "
                         "reported as the field 'weak' when N > 1; weak: only the batch-per-GPU-fixed measurement")
    ap.add_argument("--bucket-dtype", default="bf16", choices=["bf16", "f32"], help="gradient all-reduce buckets on the wire")
    ap.add_argument("--shard-optimizer", action="store_true",
                    help="N > 1: ZeRO-1 split (reduce-scatter, AdamW + EMA on 1/N of every bucket, all-gather of the bf16 weights) "
                         "instead of the default: plain all-reduce + full AdamW/EMA on every rank = the reference's DDP semantics "
                         "(main.py:347).  Opt-in: its RCCL in-place reduce-scatter / all-gather paths have only run with one rank "
                         "and over gloo, so the default multi-GPU number stays on the path whose semantics are the reference's")
    ap.add_argument("--no-shard-optimizer", action="store_true", help=argparse.SUPPRESS)      # (round-3 spelling: now the default)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-trace", action="store_true", help="skip the extra (untimed) steps that bracket GEMM launches with HIP events")
    ap.add_argument("--fp32", action="store_true", help="parity-mode kernels (not the headline number)")
    ap.add_argument("--shape-table", default=None, metavar="FILE", help="also write the traced launches per (variant, M, N, K): launches, total us, TFLOP/s")
    ap.add_argument("--graph", action="store_true", help="capture the step into one hipGraph (Trainer args.hip_graph); implies --no-trace")
    ap.add_argument("--no-graph", action="store_true", help="never capture (default: the Trainer decides -- args.hip_graph='auto' -- on one GPU: "
                                                            "a launch-bound step is captured, a GPU-bound one stays eager)")
    a = ap.parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(a.gpus)

    import vaw_amd
    vaw_amd.lib()                   # fail loudly if the HIP library is missing
    wl = WORKLOADS[a.workload]
    B = a.batch or wl["batch"]
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    scaling = a.scaling or "strong"
    if scaling == "strong" and B % world:
        raise SystemExit(f"global batch {B} does not divide over {world} GPUs")
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
    args = workload_args(wl, parallel=parallel, amp=not a.fp32, hip_graph=True if a.graph else False if (a.no_graph or parallel) else "auto")
    model, ema_model = build(vaw_amd, wl, args, device, rank)
    if a.fp32:
        model.set_compute_dtype("fp32")
    elif wl.get("fp8"):
        model.set_compute_dtype("fp8")
    shard = parallel and a.shard_optimizer and not a.no_shard_optimizer and not a.fp32
    net = vaw_amd.DistributedDataParallel(model, bucket_dtype=a.bucket_dtype, shard_optimizer=shard) if parallel else model
    opt = vaw_amd.FusedAdamW(model, lr=args.lr, betas=(0.9, 0.95), weight_decay=0.0, eps=1e-8)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=vaw_amd.get_lr_lambda(args))
    diff = vaw_amd.GaussianDiffusion(args=args, betas=vaw_amd.get_named_beta_schedule("cosine", 1000),
                                     model_mean_type=vaw_amd.ModelMeanType.EPSILON,
                                     model_var_type=vaw_amd.ModelVarType.FIXED_LARGE, loss_type=vaw_amd.LossType.MSE,
                                     rescale_timesteps=True)
    torch.manual_seed(1000 + rank)          # seed + rank: every rank draws its own t / noise (reference utils.py:62-69)

    def barrier():
        if parallel:
            dist.barrier()
        torch.cuda.synchronize()

    def run(per_gpu, first_step):
        """W untimed + K timed steps at `per_gpu` images per rank -> result fields."""
        loader = _Loader(synth_batches(per_gpu, 4, device, 123 + rank, wl))
        tr = run.tr
        if tr is None:
            tr = run.tr = vaw_amd.Trainer(args, device, net, ema_model if rank == 0 else None, opt, sched, diff, loader)
        else:
            tr.train_loader, tr.datalooper = loader, iter(loader)
        for s in range(a.warmup):
            tr.train_step(first_step + s)
        elapsed, per, losses = timed_steps(tr, first_step + a.warmup, a.steps, barrier, parallel, device)
        last = float(losses[-1])
        if not (last == last) or abs(last) > 1e4:
            raise SystemExit(f"non-finite / diverged loss {last}: the measurement is void")
        per.sort()
        return {"value": round(per_gpu * world * a.steps / elapsed, 2), "ms_per_step": round(1e3 * elapsed / a.steps, 3),
                "median_ms_per_step": round(per[len(per) // 2], 3), "per_gpu_batch": per_gpu, "global_batch": per_gpu * world,
                "last_loss": round(last, 5)}
    run.tr = None

    main_res = run(B // world if scaling == "strong" else B, 0)
    weak_res = run(B, a.warmup + a.steps) if (parallel and scaling == "strong") else None

    # roofline of the dominant kernel: extra steps AFTER the timed region (the HIP-event brackets fence the launch stream and
    # cost ~3 % of a step, so the timed steps carry none), at the batch of the headline measurement
    trace = None
    graph_timed = getattr(run.tr, "_graph", None) is not None
    if graph_timed and not a.no_trace:
        run.tr.eager_from_now_on()           # the timed steps replayed a captured graph: the traced ones launch kernel by kernel
    if not a.no_trace and rank == 0:
        trace = vaw_amd.ops.GemmTrace()
    traced_steps = 3
    if parallel or trace is not None:       # every rank runs them: the steps contain collectives
        if weak_res is not None:
            loader = _Loader(synth_batches(main_res["per_gpu_batch"], 4, device, 123 + rank, wl))
            run.tr.train_loader, run.tr.datalooper = loader, iter(loader)
        if not a.no_trace:
            run.tr.train_step(10 ** 6)
            barrier()
            vaw_amd.ops.gemm_trace = trace
            for s in range(traced_steps):
                run.tr.train_step(10 ** 6 + 1 + s)
            barrier()
            vaw_amd.ops.gemm_trace = None

    if rank == 0:
        ips, ms = main_res["value"], main_res["ms_per_step"]
        rec = {"metric": "training images/sec", "value": ips, "unit": "images/sec", "n_gpus": world,
               "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms, "median_ms_per_step": main_res["median_ms_per_step"],
               "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f32" if a.fp32 else "fp8" if wl.get("fp8") else "bf16",
               "data": "synthetic",
               "config": {"workload": f"{wl['desc']}; Trainer.train_step: q_sample + fwd + lambda-weighted MSE + bwd + "
                                      f"AdamW + EMA; global batch {main_res['global_batch']} = {main_res['per_gpu_batch']} per GPU",
                          "global_batch": main_res["global_batch"], "per_gpu_batch": main_res["per_gpu_batch"],
                          "parallelism": f"dp{world}", "weight_type": args.weight_type, "last_loss": main_res["last_loss"],
                          "grad_bucket_dtype": a.bucket_dtype if parallel else None,
                          "ddp_reserved_cus": net.reserved_cus if parallel else None, "optimizer_sharded": bool(shard),
                          "hip_graph": bool(graph_timed)}}
        if weak_res is not None:
            rec["weak"] = dict(weak_res, scaling="weak", note=f"same process, {B} images per GPU")
        if wl["gflop_per_img"]:
            rec["config"]["step_mfma_util_vs_2.5PF"] = round(ips / world * wl["gflop_per_img"] / 1e3 / BF16_MFMA_PEAK_TFLOPS, 4)
        if trace is not None:
            summ = trace.summarize()
            if a.shape_table:
                with open(a.shape_table, "w") as fh:
                    for name, shape, n, t_ms, flop in trace.by_shape():
                        fh.write(f"{name:24s} {str(shape):28s} {n / traced_steps:7.1f}/step {1e3 * t_ms / n:9.1f} us  {flop / t_ms / 1e9 if t_ms > 0 else 0:8.1f} TFLOP/s"
                                 f"  {t_ms / traced_steps:8.3f} ms/step\n")
            f8 = bool(wl.get("fp8")) and not a.fp32
            fast = {k: v for k, v in summ.items() if k.startswith("fp8_mfma" if f8 else "bf16_mfma")}
            if fast:
                flop = sum(v["flop"] for v in fast.values())
                t_ms = sum(v["ms"] for v in fast.values())
                n_l = sum(v["launches"] for v in fast.values())
                ach = flop / (t_ms * 1e-3) / 1e12
                peak = FP8_MFMA_PEAK_TFLOPS if f8 else BF16_MFMA_PEAK_TFLOPS
                traffic, src = (None, None) if a.fp32 else pmc_traffic(a.workload, main_res["per_gpu_batch"])
                rec["roofline"] = {
                    "kernel": ("fp8 MFMA GEMM family (v_mfma_scale_f32_16x16x128_f8f6f4): gemm_p8_kernel<F8> (persistent 256 x 256|192 "
                               "tiles, LDS-DMA ring) for the blocks' Linear fwd / dgrad and the grouped weight gradients; the bf16 "
                               "launches of the step are listed in by_variant") if f8 else
                              "bf16 MFMA GEMM family (v_mfma_f32_16x16x32_bf16): gemm_p8_kernel (persistent 256 x 256|192 tiles, "
                              "LDS-DMA ring, counted vmcnt) for the large Linear launches, gemm_bf16_kernel (128 x 128) for the rest; "
                              "fwd/dgrad/wgrad variants",
                    "bound": "mfma", "achieved": round(ach, 1), "peak": peak, "unit": "TFLOP/s",
                    "traffic_unit": "HBM bytes per launch (PMC, separate rocprofv3 --pmc passes of this command; a COMMITTED-PROFILE "
                                    "figure, not measured in this run: see traffic_source / traffic_collected)",
                    "frac": round(ach / peak, 4), "traffic": traffic,
                    "algorithmic_bytes_per_launch": round(sum(v["bytes"] for v in fast.values()) / n_l), "traffic_source": src,
                    "traffic_collected": pmc_collected(src),
                    "launches_per_step": n_l / traced_steps, "avg_launch_us": round(1e3 * t_ms / n_l, 2),
                    "traced_steps": traced_steps, "traced_where": "extra steps after the timed region",
                    "gemm_share_of_step": round(t_ms / traced_steps / ms, 4),
                    "by_variant": {k: {"launches_per_step": v["launches"] / traced_steps,
                                       "tflops": round(v["flop"] / (v["ms"] * 1e-3) / 1e12, 1),
                                       "avg_us": round(1e3 * v["ms"] / v["launches"], 2)} for k, v in summ.items()}}
        if world == 1 and not a.no_cpu_baseline:
            run.tr = None
            del opt, model, ema_model, net
            torch.cuda.empty_cache()
            rec["cpu_baseline"] = cpu_baseline(wl, B)
        print(json.dumps(rec), flush=True)
    if parallel:
        dist.barrier()
        vaw_amd.dist_util.cleanup_dist()


if __name__ == "__main__":
    main()
