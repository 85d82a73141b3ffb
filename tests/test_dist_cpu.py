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
