"""SURVEY §8(e) parity on the GPU: two ranks sharing the one MI355X of the test box (gloo backend; RCCL refuses two ranks
on one device) run the REAL HIP denoisers (DiT and UNet) under vaw_amd.DistributedDataParallel -- flat-buffer broadcast,
per-stage gradient buckets on the side stream, FusedAdamW -- each on half of a batch with injected per-sample t / noise; the
result must reproduce the single-rank step on the whole batch: per-sample mse and the post-step weights.  Variants: one
synchronised backward; two micro-batches per rank with the first under no_sync() (gradient accumulation, reference
tools/trainer.py:94-101); bf16 wire buckets (throughput option: looser tolerance)."""
import contextlib
import os
import socket
import sys
import traceback

import pytest
import torch
import torch.multiprocessing as mp

from conftest import REPO, base_args, perturb_

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build(vaw_amd, dev, kind):
    torch.manual_seed(21)
    if kind == "dit":
        m = vaw_amd.DiT(image_size=8, patch_size=2, in_channels=4, hidden_size=64, depth=4, num_heads=2, class_dropout_prob=0.0,
                        num_classes=10, learn_sigma=False, compute_dtype="fp32")
    else:
        m = vaw_amd.UNetModel(16, 3, 32, 3, 1, attention_resolutions=(2,), channel_mult=(1, 2), num_heads=2, num_classes=10,
                              use_scale_shift_norm=True, resblock_updown=True, use_new_attention_order=True, compute_dtype="fp32")
    perturb_(m, 31)
    return m.to(dev)


def _data(kind):
    g = torch.Generator().manual_seed(77)
    shape = (8, 4, 8, 8) if kind == "dit" else (8, 3, 16, 16)
    return (torch.randn(shape, generator=g), torch.randint(0, 10, (8,), generator=g), torch.randint(0, 1000, (8,), generator=g),
            torch.randn(shape, generator=g))


def _one_step(vaw_amd, net, model, x, y, t, noise, micro=1):
    """One optimizer step on (x, y, t, noise); micro = 2 splits it into two accumulation micro-batches, the first one under
    net.no_sync() when `net` is the data-parallel wrapper."""
    diff = vaw_amd.GaussianDiffusion(args=base_args(), betas=vaw_amd.get_named_beta_schedule("cosine", 1000),
                                     model_mean_type=vaw_amd.ModelMeanType.EPSILON, model_var_type=vaw_amd.ModelVarType.FIXED_LARGE,
                                     loss_type=vaw_amd.LossType.MSE, rescale_timesteps=True)
    opt = vaw_amd.FusedAdamW(model, lr=1e-3, betas=(0.9, 0.95), weight_decay=0.0, eps=1e-8)
    n = x.shape[0] // micro
    mses = []
    for i in range(micro):
        sl = slice(i * n, (i + 1) * n)
        ctx = net.no_sync() if (i < micro - 1 and hasattr(net, "no_sync")) else contextlib.nullcontext()
        with ctx:
            terms = diff.training_losses(net, x[sl], None, t=t[sl], model_kwargs={"y": y[sl]}, noise=noise[sl])
            (terms["loss"].mean() / micro).backward()
        mses.append(terms["mse"].detach().cpu())
    torch.cuda.synchronize()
    grads = model.flat_grads().detach().cpu().clone()      # what the wrapper is responsible for: the averaged flat gradient
    opt.step()
    if getattr(opt, "zero", None) is not None:             # sharded optimizer: every rank's f32 masters up to date again
        opt.consolidate()
        own = torch.zeros_like(grads, dtype=torch.bool)
        for lo, hi in opt._chunks:
            own[lo:hi] = True
        grads = torch.where(own, grads, torch.full_like(grads, float("nan")))     # only this rank's chunks were reduced
    torch.cuda.synchronize()
    return torch.cat(mses), model._flat.detach().cpu().clone(), grads


def _worker(rank, world, port, q, kind, variant):
    try:
        sys.path.insert(0, REPO)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
        import vaw_amd
        vaw_amd.dist_util.setup_dist(backend="gloo", device_index=0)
        dev = torch.device("cuda", 0)
        model = _build(vaw_amd, dev, kind)
        if rank != 0:
            model.ensure_flat()
            with torch.no_grad():
                model._flat.add_(0.5)                      # ranks start different: the wrapper's broadcast must equalise them
        net = vaw_amd.DistributedDataParallel(model, bucket_dtype="bf16" if variant == "bf16_buckets" else "f32",
                                              shard_optimizer=variant == "zero")
        x, y, t, noise = (v[4 * rank:4 * rank + 4].to(dev) for v in _data(kind))
        mse, flat, grads = _one_step(vaw_amd, net, model, x, y, t, noise, micro=2 if variant == "no_sync" else 1)
        q.put((rank, mse.numpy(), flat.numpy(), grads.numpy(), None))      # by value: the worker exits before the parent reads
        vaw_amd.dist_util.cleanup_dist()
    except Exception:
        q.put((rank, None, None, None, traceback.format_exc()))


@pytest.mark.parametrize("kind,variant", [("dit", "sync"), ("dit", "no_sync"), ("dit", "bf16_buckets"), ("unet", "sync"), ("unet", "no_sync"),
                                          ("dit", "zero"), ("unet", "zero")])
def test_two_rank_step_reproduces_single_rank_step(kind, variant):
    import vaw_amd
    dev = torch.device("cuda", 0)
    model = _build(vaw_amd, dev, kind)
    x, y, t, noise = (v.to(dev) for v in _data(kind))
    mse1, flat1, grads1 = _one_step(vaw_amd, model, model, x, y, t, noise)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, kind, variant)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        rank, mse, flat, grads, err = q.get(timeout=300)
        assert err is None, err
        res[rank] = (torch.from_numpy(mse), torch.from_numpy(flat), torch.from_numpy(grads))
    for p in procs:
        p.join(timeout=60)
    # per-sample objective: each rank's half equals the corresponding half of the single-rank batch
    torch.testing.assert_close(torch.cat([res[0][0], res[1][0]]), mse1, rtol=1e-5, atol=1e-7)
    # the averaged flat gradient itself: bitwise equal on both ranks, and equal to the single-rank gradient of the whole batch
    # within 1e-6 of the tensor's rms (f32 buckets: only the order of summation differs; bf16 buckets round each rank's
    # contribution to 8 bits on the wire: 2^-8 relative per element)
    if variant == "zero":      # reduce-scatter: each rank holds the averaged gradient of its own chunks (NaN elsewhere); together all
        g0, g1 = res[0][2], res[1][2]
        assert bool((g0.isnan() ^ g1.isnan()).all())
        res[0] = (res[0][0], res[0][1], torch.where(g0.isnan(), g1, g0))
    else:
        assert torch.equal(res[0][2], res[1][2])
    rms = float(grads1.pow(2).mean().sqrt())
    gd = (res[0][2] - grads1).abs()
    if variant == "bf16_buckets":       # each rank's share and the wire sum are rounded to 8 significant bits
        tol = 2.0 ** -7 * grads1.abs() + 2.0 ** -6 * rms
    else:                               # order of summation only: 1e-5 of the tensor's rms (+ f32 rounding of large elements) ...
        tol = 1e-5 * rms + 2e-6 * grads1.abs()
    # ... for all but a handful of elements whose sum cancels (f32 rounding scales with sum |terms|, not with the result.
    # Measured against 1e-6 of the rms: 6 of 1.1 M elements of the DiT beyond it, the worst 1.7e-5 of the rms; 467 of 0.8 M of
    # the UNet, the worst 1.1e-5); everything stays within 1e-4 of the rms
    beyond = gd > tol
    assert float(beyond.float().mean()) < 1e-4, (float(gd.max()), rms, int(beyond.sum()))
    if variant != "bf16_buckets":
        assert float(gd.max()) <= 1e-4 * rms, (float(gd.max()), rms)
    # averaged gradients -> identical AdamW update on both ranks, equal to the single-rank update (reduction-order tolerance)
    assert torch.equal(res[0][1], res[1][1])
    if variant == "bf16_buckets":       # the sum is taken in bf16 on the wire: AdamW's first step is +-lr per element whatever the
        torch.testing.assert_close(res[0][1], flat1, rtol=0, atol=2.1e-3)      # magnitude, so only sign flips of ~0 gradients differ
        assert float((res[0][1] - flat1).abs().gt(1e-5).float().mean()) < 0.02
    else:
        # f32 buckets: equal up to reduction order.  AdamW's first step moves an element by lr * g / (|g| + eps): where the
        # gradient itself is of the order of eps = 1e-8 (a handful of UNet weights) the order of summation shows, bounded by 2 lr
        d = (res[0][1] - flat1).abs()
        off = d > 2e-6 + 1e-4 * flat1.abs()
        assert float(off.float().mean()) < 1e-4 and float(d.max()) < 2.1e-3, (int(off.sum()), float(d.max()))


def _zero_bf16_worker(rank, world, port, q, shard):
    try:
        sys.path.insert(0, REPO)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
        import vaw_amd
        vaw_amd.dist_util.setup_dist(backend="gloo", device_index=0)
        dev = torch.device("cuda", 0)
        model = _build(vaw_amd, dev, "dit")
        model.set_compute_dtype("bf16")
        net = vaw_amd.DistributedDataParallel(model, shard_optimizer=shard)
        diff = vaw_amd.GaussianDiffusion(args=base_args(), betas=vaw_amd.get_named_beta_schedule("cosine", 1000),
                                         model_mean_type=vaw_amd.ModelMeanType.EPSILON, model_var_type=vaw_amd.ModelVarType.FIXED_LARGE,
                                         loss_type=vaw_amd.LossType.MSE, rescale_timesteps=True)
        opt = vaw_amd.FusedAdamW(model, lr=1e-2, betas=(0.9, 0.95), weight_decay=0.0, eps=1e-8)
        x, y, t, noise = (v[4 * rank:4 * rank + 4].to(dev) for v in _data("dit"))
        mses = []
        for _ in range(3):
            terms = diff.training_losses(net, x, None, t=t, model_kwargs={"y": y}, noise=noise)
            terms["loss"].mean().backward()
            opt.step()
            opt.zero_grad()
            mses.append(terms["mse"].detach().cpu().numpy())
        if shard:
            # the f32 masters of the other rank's chunks are stale now: everything that reads or re-derives from them refuses
            # (a silent stale mix was the alternative) until the collective consolidate() has run on every rank
            for what, fn in (("state_dict", lambda: model.state_dict()), ("set_compute_dtype", lambda: model.set_compute_dtype("fp32"))):
                try:
                    fn()
                    raise AssertionError(f"{what} on stale masters did not raise")
                except RuntimeError as e:
                    assert "consolidate" in str(e), (what, str(e))
            opt.consolidate()
            model.state_dict()
        torch.cuda.synchronize()
        q.put((rank, mses, model._flat.detach().cpu().numpy(), None))
        vaw_amd.dist_util.cleanup_dist()
    except Exception:
        q.put((rank, None, None, traceback.format_exc()))


def test_sharded_optimizer_bf16_mode_matches_unsharded_bitwise():
    """bf16 compute: the sharded optimizer all-gathers the bf16 shadow plus the few parameters the kernels read as f32 (biases,
    embedding rows).  Three steps sharded == three steps with all-reduce + full update, bit for bit (losses of every step on
    both ranks, and the consolidated f32 masters)."""
    out = {}
    for shard in (False, True):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_zero_bf16_worker, args=(r, 2, port, q, shard)) for r in range(2)]
        for p in procs:
            p.start()
        res = {}
        for _ in range(2):
            rank, mses, flat, err = q.get(timeout=300)
            assert err is None, err
            res[rank] = (mses, flat)
        for p in procs:
            p.join(timeout=60)
        out[shard] = res
    for rank in (0, 1):
        for a, b in zip(out[False][rank][0], out[True][rank][0]):
            assert (a == b).all(), (rank, a, b)
        assert (out[False][rank][1] == out[True][rank][1]).all()


def _direct_world1_worker(q):
    """Runs in a child process: the library's RCCL communicator lives for the life of a process."""
    try:
        import ctypes
        sys.path.insert(0, REPO)
        import vaw_amd  # noqa: F401
        from vaw_amd import _lib as L
        lib = L.lib()
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(0)
        buf = torch.ones(1024, device=dev)
        # before vaw_comm_init every data call is refused, loudly
        assert lib.vaw_comm_world() == 0
        assert lib.vaw_allreduce_bucket_start(L.ptr(buf), buf.numel(), L.F32, L.stream_ptr()) != 0
        assert b"no communicator" in lib.vaw_last_error_string()
        ident = (ctypes.c_ubyte * 128)()
        L.check(lib.vaw_comm_unique_id(ident), "unique_id")
        L.check(lib.vaw_comm_init(ident, 0, 1), "comm_init")
        assert lib.vaw_comm_world() == 1
        assert lib.vaw_comm_init(ident, 0, 1) != 0                     # one communicator per process
        g = torch.Generator().manual_seed(3)
        for dt, code in ((torch.float32, L.F32), (torch.bfloat16, L.BF16)):
            n = 3 * 1024 * 1024 + 8
            want = torch.randn(n, generator=g).to(dt)
            x = torch.zeros(n, device=dev, dtype=dt)
            big = torch.randn(4096, 4096, device=dev)
            for _ in range(4):
                big = big @ big * 1e-3                                   # keeps the compute stream busy in front of the bucket's producer
            x.copy_(want.to(dev), non_blocking=True)                    # the "last wgrad kernel" of the bucket, on the current stream
            L.check(lib.vaw_allreduce_bucket_start(L.ptr(x), n, code, L.stream_ptr()), "allreduce")
            L.check(lib.vaw_allreduce_bucket_wait(L.stream_ptr()), "wait")
            got = x.clone()                                             # current stream: ordered behind the collective by the wait
            torch.cuda.synchronize()
            assert torch.equal(got.cpu(), want), dt                     # mean over one rank: the bucket itself, to the bit
            y = x.clone()
            L.check(lib.vaw_reduce_scatter_bucket_start(L.ptr(y), n, code, L.stream_ptr()), "reduce_scatter")
            L.check(lib.vaw_allgather_bucket_start(L.ptr(y), n, code, L.stream_ptr()), "allgather")
            L.check(lib.vaw_allreduce_bucket_wait(L.stream_ptr()), "wait")
            torch.cuda.synchronize()
            assert torch.equal(y.cpu(), want), dt
        assert lib.vaw_reduce_scatter_bucket_start(None, 8, L.F32, L.stream_ptr()) != 0
        L.check(lib.vaw_comm_destroy(), "comm_destroy")
        assert lib.vaw_comm_world() == 0
        q.put("ok")
    except Exception:
        q.put(traceback.format_exc())


@pytest.mark.gpu
def test_direct_rccl_bucket_collectives_on_one_rank():
    """vaw_allreduce_bucket_start / _wait (csrc/collective.hip) on the only world size a one-GPU box offers: the
    communicator comes up from its own unique id, a bucket produced on the compute stream is reduced on the library's side stream
    behind it and read back behind the wait (mean over one rank = the bucket, bit for bit, f32 and bf16 wire types), the ZeRO pair
    reduce-scatter + all-gather round-trips, and every call without a communicator or with a bad bucket is refused with a message."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_direct_world1_worker, args=(q,))
    p.start()
    p.join(300)
    assert not p.is_alive()
    res = q.get(timeout=10)
    assert res == "ok", res
