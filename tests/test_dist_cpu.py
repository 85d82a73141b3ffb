"""N>1 path on CPU: two gloo processes exercise the data-parallel wrapper (parameter broadcast, bucketed
gradient all-reduce per backward stage, no_sync, 'module.' state_dict prefix) and the loss-aware sampler's
gather.  The wrapper is backend-agnostic; on the MI355X box the same code runs over RCCL ("nccl")."""
import os
import socket
import sys
import traceback
from types import SimpleNamespace

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from conftest import REPO


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    try:
        sys.path.insert(0, REPO)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                          LOCAL_RANK=str(rank))
        import vaw_amd
        from vaw_amd.flat import FlatModule

        class Toy(FlatModule):
            """CPU stand-in for a HIP denoiser: same flat-storage + backward-stage protocol, no kernels."""

            def __init__(self, seed):
                super().__init__()
                torch.manual_seed(seed)
                self.a = nn.Linear(5, 7)
                self.b = nn.Linear(7, 3)
                self.grad_ready_hook = None

            def grad_stage_bounds(self):
                self.ensure_flat()
                cut = self._flat_offsets["b.weight"][0]
                return {1: (cut, self._flat_n_train), 0: (0, cut)}

            def fake_backward(self, value):
                g = self.flat_grads()
                live = self.grads_live()
                cut = self._flat_offsets["b.weight"][0]
                for lo, hi, stage in ((cut, self._flat_n_train, 1), (0, cut, 0)):
                    g[lo:hi] = (g[lo:hi] if live else 0) + value
                    if stage == 0:
                        self.attach_grads()
                    if self.grad_ready_hook:
                        self.grad_ready_hook(stage)

        vaw_amd.dist_util.setup_dist()
        assert dist.get_backend() == "gloo" and dist.get_world_size() == world
        assert vaw_amd.dist_util.is_main_process() == (rank == 0)
        m = Toy(seed=100 + rank)                      # ranks start DIFFERENT; the wrapper must equalise them
        w0 = m.a.weight.detach().clone()
        ddp = vaw_amd.DistributedDataParallel(m)
        gathered = [torch.zeros_like(w0) for _ in range(world)]
        dist.all_gather(gathered, m.a.weight.detach().clone())
        assert all(torch.equal(gathered[0], t) for t in gathered), "parameters not broadcast from rank 0"
        assert all(k.startswith("module.") for k in ddp.state_dict())
        # one synchronised backward: every rank contributes (rank+1) -> mean over ranks
        m.fake_backward(float(rank + 1))
        mean = sum(range(1, world + 1)) / world
        assert torch.allclose(m.a.weight.grad, torch.full_like(m.a.weight, mean))
        assert torch.allclose(m.b.bias.grad, torch.full_like(m.b.bias, mean))
        pad = m.flat_grads().clone()
        # accumulation micro-step under no_sync stays local, the following synchronised one reduces the sum
        m.zero_grad_flat()
        with ddp.no_sync():
            m.fake_backward(float(rank + 1))
        assert torch.allclose(m.a.weight.grad, torch.full_like(m.a.weight, float(rank + 1)))
        m.fake_backward(10.0)
        assert torch.allclose(m.a.weight.grad, torch.full_like(m.a.weight, mean + 10.0))
        # bf16 wire buckets: cast -> all-reduce at half the bytes -> widen back (values exact in bf16 here)
        ddp16 = vaw_amd.DistributedDataParallel(m, broadcast=False, bucket_dtype="bf16")
        m.zero_grad_flat()
        m.fake_backward(float(rank + 1))
        assert m.flat_grads().dtype == torch.float32
        assert torch.allclose(m.a.weight.grad, torch.full_like(m.a.weight, mean))
        assert torch.allclose(m.b.bias.grad, torch.full_like(m.b.bias, mean))
        with pytest.raises(ValueError):
            vaw_amd.DistributedDataParallel(m, broadcast=False, bucket_dtype="fp8")
        # the library's own RCCL communicator needs a GPU and an RCCL process group: refused here, not silently torch.distributed
        with pytest.raises(ValueError, match="RCCL"):
            vaw_amd.DistributedDataParallel(m, broadcast=False, collectives="direct")
        with pytest.raises(ValueError, match="collectives"):
            vaw_amd.DistributedDataParallel(m, broadcast=False, collectives="mpi")
        del ddp16
        # loss-aware sampler: ranks hold different numbers of (t, loss) pairs; histories must end identical
        s = vaw_amd.create_named_schedule_sampler("loss-second-moment", SimpleNamespace(num_timesteps=6))
        ts = torch.tensor([rank, 5]) if rank == 0 else torch.tensor([2, 3, 4])
        ls = ts.float() * 0.5 + rank
        s.update_with_local_losses(ts, ls)
        hist = torch.from_numpy(s._ring.copy())
        gh = [torch.zeros_like(hist) for _ in range(world)]
        dist.all_gather(gh, hist)
        assert all(torch.equal(gh[0], h) for h in gh)
        assert int(s._seen.sum()) == 5
        vaw_amd.dist_util.dist_barrier()
        vaw_amd.dist_util.cleanup_dist()
        q.put((rank, "ok"))
    except Exception:
        q.put((rank, traceback.format_exc()))


def test_data_parallel_wrapper_two_gloo_ranks():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in results:
        assert msg == "ok", f"rank {rank}:\n{msg}"


# ---- sharded optimizer (ZeRO-1) and the real DiT bucket layout, incl. the early adaLN bucket ---------------------------------
def _adamw_torch(p, g, m, v, ema, shadow, lr, beta1, beta2, eps, wd, step, ema_decay, sumsq_t, clip, zero_grad, hyper=None):
    """torch statement of vaw_adamw_ema_step for the CPU ranks of this test (the product has no CPU path: test stand-in only).
    One IEEE operation per torch call (no alpha= / value= forms, which may or may not contract into an FMA depending on how a
    slice falls on the vector width): the result of an element then cannot depend on how the buffer is cut into chunks."""
    scale = 1.0
    if clip:
        scale = min(1.0, float(clip) / (float(sumsq_t.sqrt()) + 1e-6))
    gg = g * scale
    p.mul_(1 - lr * wd)
    m.mul_(beta1).add_(gg * (1 - beta1))
    v.mul_(beta2).add_((gg * gg) * (1 - beta2))
    denom = (v.sqrt() / (1 - beta2 ** step) ** 0.5) + eps
    p.sub_((m / denom) * (lr / (1 - beta1 ** step)))
    if ema is not None:
        ema.mul_(ema_decay).add_(p * (1 - ema_decay))
    if shadow is not None:
        shadow.copy_(p)


def _sumsq_torch(g, out, accumulate=False):
    out.copy_((out if accumulate else 0) + g.double().pow(2).sum().float())


def _zero_worker(rank, world, port, q):
    try:
        sys.path.insert(0, REPO)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
        import vaw_amd
        from vaw_amd import ops
        ops.adamw_ema_step, ops.sumsq = _adamw_torch, _sumsq_torch          # CPU stand-ins for the HIP kernels of the update
        ops.ema_update = lambda ema, src, decay: ema.mul_(decay).add_(src * (1 - decay))
        vaw_amd.dist_util.setup_dist()

        def build():
            torch.manual_seed(5)
            m = vaw_amd.DiT(image_size=8, patch_size=2, in_channels=4, hidden_size=64, depth=4, num_heads=2, class_dropout_prob=0.0,
                            num_classes=10, learn_sigma=False, compute_dtype="fp32")
            with torch.no_grad():
                for p in m.parameters():
                    if p.requires_grad:
                        p.add_(torch.randn(p.shape) * 0.05)
            return m

        def fake_backward(m, step):
            """Every stage of the real DiT backward fires in its order (head, blocks L-1..0 with the early adaLN bucket half-way,
            embedders); gradients are small integers that depend on rank, element and step: sums over ranks are exact."""
            g = m.flat_grads()
            n = g.numel()
            g.copy_(((torch.arange(n) % 7) - 3 + rank + step).float())
            m.attach_grads()
            hook = m.grad_ready_hook
            L = m.depth
            hook(L + 1)
            for l in reversed(range(L)):
                hook(l + 1)
                if l == m._ada_split_block():
                    hook("ada_hi")
            hook(0)

        def run(shard, clip):
            m = build()
            ddp = vaw_amd.DistributedDataParallel(m, shard_optimizer=shard)
            # the buckets partition the trainable range exactly, early adaLN bucket included
            rngs = ddp.bucket_ranges()
            assert rngs[0][0] == 0 and rngs[-1][1] == m._flat_n_train and all(a[1] == b[0] for a, b in zip(rngs, rngs[1:]))
            assert any(isinstance(v, list) for v in ddp._ranges.values()), "expected the split adaLN stage"
            # the sharded optimizer's all-gathers go out in FORWARD order (embedders + adaLN first, blocks ascending, head last)
            # and cover exactly the buckets
            st = ddp.gather_stages()
            assert [k for k, _ in st] == [0, "ada_hi"] + list(range(1, m.depth + 2))
            assert sorted(r for _, rs in st for r in rs) == rngs
            opt = vaw_amd.FusedAdamW(m, lr=1e-2, betas=(0.9, 0.95), weight_decay=0.01)
            ema_model = build() if rank == 0 else None
            if shard:
                opt.attach_ema_sharded(0.9, ema_model)
            elif rank == 0:
                opt.attach_ema(ema_model, 0.9)
            opt.max_grad_norm = clip
            for step in range(3):
                fake_backward(m, step)
                opt.step()
                opt.zero_grad()
            if shard:
                opt.consolidate()
                opt.consolidate_ema(ema_model)
            sd = opt.state_dict()
            return m._flat.clone(), (ema_model._flat.clone() if rank == 0 else None), sd

        for clip in (None, 2.5):
            ref_p, ref_e, ref_sd = run(False, clip)
            got_p, got_e, got_sd = run(True, clip)
            assert torch.equal(ref_p, got_p), f"clip={clip}: parameters differ by {float((ref_p - got_p).abs().max())}"
            if rank == 0:
                assert torch.equal(ref_e, got_e), f"clip={clip}: EMA differs by {float((ref_e - got_e).abs().max())}"
            for i, st in ref_sd["state"].items():
                assert torch.equal(st["exp_avg"], got_sd["state"][i]["exp_avg"]) and torch.equal(st["exp_avg_sq"], got_sd["state"][i]["exp_avg_sq"])
        # a sharded optimizer restored from a full state dict continues like the unsharded one
        m = build()
        ddp = vaw_amd.DistributedDataParallel(m, shard_optimizer=True)
        opt = vaw_amd.FusedAdamW(m, lr=1e-2, betas=(0.9, 0.95), weight_decay=0.01)
        opt.load_state_dict(ref_sd)
        assert opt.step_count == 3 and opt.exp_avg.numel() * world == m._flat_n_train
        # misuse guards: an optimizer built BEFORE the wrap (zero captured as None) must not run its full update on the
        # reduce-scattered gradients, and a Trainer must not accept a non-sharded optimizer on such a model
        m = build()
        early = vaw_amd.FusedAdamW(m, lr=1e-2)
        ddp = vaw_amd.DistributedDataParallel(m, shard_optimizer=True)
        fake_backward(m, 0)
        try:
            early.step()
            raise AssertionError("FusedAdamW built before the shard_optimizer wrap stepped")
        except RuntimeError as e:
            assert "BEFORE" in str(e), str(e)
        from conftest import base_args
        diff = vaw_amd.GaussianDiffusion(args=base_args(), betas=vaw_amd.get_named_beta_schedule("cosine", 1000),
                                         model_mean_type=vaw_amd.ModelMeanType.EPSILON, model_var_type=vaw_amd.ModelVarType.FIXED_LARGE,
                                         loss_type=vaw_amd.LossType.MSE, rescale_timesteps=True)
        for bad in (early, torch.optim.AdamW(m.parameters(), lr=1e-2)):
            try:
                vaw_amd.Trainer(base_args(parallel=True, amp=False), torch.device("cpu"), ddp, None, bad, None, diff, [(torch.zeros(1), torch.zeros(1))])
                raise AssertionError("Trainer accepted a non-sharded optimizer on a shard_optimizer model")
            except ValueError as e:
                assert "shard_optimizer" in str(e), str(e)
        # the stale-master flag lives on the module: state_dict() refuses until the collective consolidate() has run
        late = vaw_amd.FusedAdamW(m, lr=1e-2)
        late.master_stale = m._master_stale = True
        try:
            m.state_dict()
            raise AssertionError("state_dict() on stale masters did not raise")
        except RuntimeError as e:
            assert "consolidate" in str(e)
        late.consolidate()
        assert not m._master_stale and m.state_dict()
        vaw_amd.dist_util.dist_barrier()
        vaw_amd.dist_util.cleanup_dist()
        q.put((rank, "ok"))
    except Exception:
        q.put((rank, traceback.format_exc()))


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_optimizer_equals_unsharded_on_the_dit_bucket_layout(world):
    """ZeRO-1 (reduce-scatter -> AdamW + EMA on 1 / world of every bucket -> all-gather) reproduces the all-reduce + full
    update bit for bit (integer gradients), with the REAL DiT stage bounds: head, blocks, the early adaLN bucket, embedders."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_zero_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in results:
        assert msg == "ok", f"rank {rank}:\n{msg}"
