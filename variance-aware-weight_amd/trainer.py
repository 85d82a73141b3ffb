"""One optimizer step of variance-aware-weighted diffusion training; same constructor and `train_step(step)`
as the reference's tools/trainer.py:28-150, plus `ema` :12-18 and `sample_from_latent` :21-25.

Differences that do not change results:
  * `args.amp=True` selects the bf16 MFMA kernels (no GradScaler: bf16 keeps the f32 exponent range);
    `args.amp=False` selects the f32 parity kernels.
  * with `vaw_amd.FusedAdamW`, clip + AdamW + EMA + bf16 shadow refresh are one kernel; any other
    torch optimizer is driven exactly as the reference drives it.
  * the two `.item()` host syncs per micro-step become one per step (or none: `args.defer_loss_sync=True`
    makes train_step return a 0-dim device tensor).
  * `args.hip_graph=True` (single GPU, FusedAdamW, grad_accumulation 1): after two eager steps the whole step -- latent
    sampling, q_sample, forward, loss, backward, clip, AdamW+EMA, zero_grad: several hundred launches -- is captured
    into ONE hipGraph and replayed; per step the host only copies the next batch into the static input buffers and
    refreshes three optimizer scalars in device memory.  For launch-bound configurations (small models / batches).
"""
from contextlib import nullcontext

import torch
import torch.nn as nn

from . import dist_util, ops
from .flat import FlatModule
from .optim import FusedAdamW


def ema(source, target, decay):
    """target = target*decay + source*(1-decay) over EVERY state_dict entry (buffers and frozen params too)."""
    s, t = getattr(source, "module", source), getattr(target, "module", target)
    with torch.no_grad():
        if isinstance(s, FlatModule) and isinstance(t, FlatModule):
            s.ensure_flat(); t.ensure_flat()
            if s._flat_offsets == t._flat_offsets and s._flat.is_cuda and not list(s.buffers()):
                ops.ema_update(t._flat, s._flat, decay)
                t.mark_weights_changed()
                return
        src, dst = source.state_dict(), target.state_dict()
        for k in src:
            dst[k].data.copy_(dst[k].data * decay + src[k].data * (1 - decay))
        if isinstance(t, FlatModule) and getattr(t, "_flat", None) is not None:
            t.mark_weights_changed()           # `.data` writes do not move parameter versions either


def sample_from_latent(latent, latent_scale=1.0, cpu_rng=False):
    """latent = cat[mean, std] of the VAE posterior -> scaled sample.  cpu_rng draws from the CPU generator
    (the stream the CPU reference consumes) instead of the device generator."""
    mean, std = torch.chunk(latent, 2, dim=1)
    eps = torch.randn(mean.shape).to(mean.device) if cpu_rng else torch.randn_like(mean)
    return (mean + std * eps) * latent_scale


class Trainer:
    def __init__(self, args, device, model, ema_model, optimizer, scheduler, diffusion, train_loader, pbar=None):
        if getattr(args, "learn_align", False):
            raise NotImplementedError("learn_align needs network-fetched teacher encoders: out of scope (SURVEY.md §2.1 row 12)")
        self.args, self.device = args, device
        self.model, self.ema_model = model, ema_model
        self.optimizer, self.scheduler, self.diffusion = optimizer, scheduler, diffusion
        self.train_loader = train_loader
        self.datalooper = iter(train_loader)
        self.pbar = pbar
        self.scaler = None
        inner = getattr(model, "module", model)
        if hasattr(inner, "set_compute_dtype"):
            # --amp (main.py:104) selects reduced-precision GEMMs: bf16, or fp8 where the model was built / configured for it
            # (compute_dtype="fp8" or args.amp_dtype = "fp8", an extension: the reference has bf16 autocast only)
            low = getattr(args, "amp_dtype", None) or (inner.compute_dtype if inner.compute_dtype in ("bf16", "fp8") else "bf16")
            want = low if args.amp else "fp32"
            if inner.compute_dtype != want:
                inner.set_compute_dtype(want)
        if hasattr(inner, "host_dropout_rng"):
            inner.host_dropout_rng = bool(getattr(args, "cpu_rng", False))      # dropout masks from the CPU stream in parity runs
        self._fused = isinstance(optimizer, FusedAdamW)
        self._zero = self._fused and getattr(optimizer, "zero", None) is not None
        if getattr(inner, "_zero", None) is not None and not self._zero:
            # gradients are reduce-scattered (shard_optimizer=True): only a FusedAdamW built AFTER the wrap updates its own chunks
            # and gathers the weights; any other optimizer would step on mostly un-reduced gradients and the ranks would diverge
            raise ValueError("the model was wrapped with shard_optimizer=True: it needs a vaw_amd.FusedAdamW built after the wrap "
                             f"(got {type(optimizer).__name__}{' built before the wrap' if self._fused else ''})")
        if self._zero:
            # sharded optimizer: every rank averages its own chunks; rank 0's ema_model receives them on consolidate()
            if getattr(args, "ema_decay", None) is not None:
                optimizer.attach_ema_sharded(args.ema_decay, ema_model)
        elif self._fused and ema_model is not None and dist_util.is_main_process():
            optimizer.attach_ema(ema_model, args.ema_decay)
        self.last_mse = None
        self._cpu_rng = bool(getattr(args, "cpu_rng", False))
        # optional importance sampling of t (resample.py exists in the reference but its Trainer never
        # calls it).  args.schedule_sampler = "loss-second-moment": t ~ sampler, loss = mean(w_t * loss_t) with the
        # sampler's 1/(T p_t) weights (guided-diffusion's TrainLoop), history updated from the per-sample losses
        self.schedule_sampler = None
        name = getattr(args, "schedule_sampler", None)
        if name and name != "uniform":
            from .resample import create_named_schedule_sampler
            self.schedule_sampler = create_named_schedule_sampler(name, diffusion)
            if getattr(args, "hip_graph", False):
                raise ValueError("args.schedule_sampler updates its history on the host every step: not with args.hip_graph")
        # hipGraph mode
        self._graph, self._graph_calls, self._gin, self._gout = None, 0, None, None
        # args.hip_graph: True | False | "auto".  "auto" captures the step only when it is LAUNCH-bound (the host needs about as long
        # to enqueue the step's kernels as the GPU needs to run them: the CIFAR-shaped UNet's ~800 five-microsecond kernels,
        # 10.6 -> 8.1 ms) and the admission check below passes; a GPU-bound step (DiT-B/4 at any batch: +-0.5 %) stays eager
        hg = getattr(args, "hip_graph", False)
        self._graph_auto = hg == "auto"
        self._use_graph = bool(hg)
        if self._use_graph:
            why = ("needs vaw_amd.FusedAdamW" if not self._fused else "not with DDP (args.parallel)" if args.parallel else
                   "not with grad_accumulation > 1" if max(1, args.grad_accumulation) > 1 else
                   "not with args.cpu_rng" if self._cpu_rng else "needs a CUDA device" if torch.device(device).type != "cuda" else
                   "the model draws its label-drop mask on the host every step (UNet drop_label_prob > 0)"
                   if getattr(inner, "host_rng_in_forward", False) else None)
            if why and self._graph_auto:
                self._use_graph = False              # "auto" quietly stays eager where a graph cannot be taken
            elif why:
                raise ValueError(f"args.hip_graph: {why}")
            else:
                optimizer.enable_device_hyper()

    def _get_next_batch(self):
        try:
            images, labels = next(self.datalooper)
        except StopIteration:
            self.datalooper = iter(self.train_loader)
            return self._get_next_batch()
        return (images.to(self.device, non_blocking=True),
                labels.to(self.device, non_blocking=True) if self.args.class_cond else None)

    def _compute_loss(self, images, labels, features):
        model_kwargs = {"y": labels} if self.args.class_cond else {}
        if self._cpu_rng:
            # parity runs: noise, then t, from the CPU generator, in the reference's order
            # (tools/gaussian_diffusion.py:849-852); both are accepted keyword arguments there too
            noise = torch.randn(images.shape).to(images.device)
            t = torch.randint(0, self.diffusion.num_timesteps, (images.shape[0],)).to(images.device)
            return self.diffusion.training_losses(self.model, images, features, t=t, model_kwargs=model_kwargs, noise=noise)
        if self.schedule_sampler is not None:
            t, w = self.schedule_sampler.sample(images.shape[0], images.device)
            terms = self.diffusion.training_losses(self.model, images, features, t=t, model_kwargs=model_kwargs)
            self.schedule_sampler.update_with_local_losses(t, terms["loss"].detach())
            terms = dict(terms)
            terms["loss"] = terms["loss"] * w
            return terms
        return self.diffusion.training_losses(self.model, images, features, model_kwargs=model_kwargs)

    def _apply_gradient_clipping(self):
        if self.args.grad_clip:
            if self._fused:
                self.optimizer.max_grad_norm = float(self.args.grad_clip)
            else:
                nn.utils.clip_grad_norm_(self.model.parameters(), self.args.grad_clip)

    def consolidate(self):
        """Sharded optimizer (ZeRO-1): gather the f32 master weights on every rank and the sharded EMA into rank 0's ema_model
        -- call on EVERY rank before sampling / evaluating / saving.  No-op otherwise."""
        if self._zero:
            self.optimizer.consolidate()
            if getattr(self.optimizer, "_ema_shard", None) is not None:
                self.optimizer.consolidate_ema(self.ema_model)

    def _update_ema(self):
        if self._zero:
            return            # done inside the sharded update, on every rank
        if dist_util.is_main_process() and self.ema_model is not None:
            if self._fused and getattr(self.optimizer, "ema_done_in_step", False):
                return
            ema(self.model, self.ema_model, self.args.ema_decay)

    def _graph_body(self, images, labels):
        """Everything of one step that runs on the GPU, on static inputs; returns (loss, mse) device scalars."""
        a = self.args
        if a.in_chans == 4:
            images = sample_from_latent(images, a.latent_scale, False)
        loss_dict = self._compute_loss(images, labels, None)
        loss = loss_dict["loss"].mean()
        loss.backward()
        self._apply_gradient_clipping()
        self.optimizer.step()
        self.optimizer.zero_grad()
        # pure KL / RESCALED_KL objectives carry no "mse" term (reference tools/trainer.py:116 guards the same way)
        return loss.detach(), (loss_dict["mse"].detach().mean() if "mse" in loss_dict else torch.zeros_like(loss.detach()))

    def eager_from_now_on(self):
        """Drop a captured graph: later steps launch their kernels one by one again (bench.py brackets them with HIP events)."""
        if self._use_graph:
            self._graph, self._gin, self._gout = None, None, None
            self._graph_calls = -(1 << 60)

    def _train_step_graph(self, step):
        a = self.args
        self.model.train()
        images, labels = self._get_next_batch()
        self.optimizer.prepare_step()                     # host: step count, {lr, bc1, bc2} -> device
        self._graph_calls += 1
        if self._graph is None and self._graph_calls <= 2:
            if self._graph_auto and self._graph_calls == 2:
                # second warm-up step: host enqueue time against GPU time decides whether replaying a graph can pay
                import time
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                e0.record()
                total, mse = self._graph_body(images, labels)
                e1.record()
                host_ms = 1e3 * (time.perf_counter() - t0)
                torch.cuda.synchronize()
                self.graph_decision = dict(host_ms=host_ms, gpu_ms=e0.elapsed_time(e1))
                if host_ms < 0.7 * self.graph_decision["gpu_ms"]:
                    self._graph_calls = -(1 << 60)       # GPU-bound: never capture (the eager body keeps running)
            else:
                total, mse = self._graph_body(images, labels)   # eager warm-up: lazy initialisation, workspaces, scratch
        elif self._graph_calls < 0:
            total, mse = self._graph_body(images, labels)
        else:
            if self._graph is None:
                self._gin = (images.clone(), None if labels is None else labels.clone())
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._gout = self._graph_body(*self._gin)
                self._graph = g
            else:
                self._gin[0].copy_(images, non_blocking=True)
                if labels is not None:
                    self._gin[1].copy_(labels, non_blocking=True)
            self._graph.replay()
            total, mse = self._gout
        self.scheduler.step()
        if dist_util.is_main_process():
            self._update_ema()
        self.last_mse = mse
        if dist_util.is_main_process() and self.pbar is not None:
            self.pbar.update(1)
        if getattr(a, "defer_loss_sync", False):
            return total
        total_f = float(total.item())
        if dist_util.is_main_process() and self.pbar is not None:
            self.pbar.set_postfix(mse=f"{float(mse.item()):.4f}")
        return total_f

    def train_step(self, step):
        if self._use_graph:
            return self._train_step_graph(step)
        a = self.args
        self.model.train()
        if a.parallel:
            self.train_loader.sampler.set_epoch(step)
        accum = max(1, a.grad_accumulation)
        total, mse_avg = None, None
        for i in range(accum):
            images, labels = self._get_next_batch()
            if a.in_chans == 4:
                images = sample_from_latent(images, a.latent_scale, self._cpu_rng)
            if a.parallel and accum > 1 and i < accum - 1:
                ctx = self.model.no_sync()
            else:
                ctx = nullcontext()
            with ctx:
                loss_dict = self._compute_loss(images, labels, None)
                loss = loss_dict["loss"].mean() / accum
                loss.backward()
            ld = loss.detach()
            md = loss_dict["mse"].detach().mean() / accum if "mse" in loss_dict else torch.zeros_like(ld)
            total = ld if total is None else total + ld
            mse_avg = md if mse_avg is None else mse_avg + md
            if (i + 1) % accum == 0:
                self._apply_gradient_clipping()
                self.optimizer.step()
                self.optimizer.zero_grad()
        self.scheduler.step()
        if dist_util.is_main_process():
            self._update_ema()
        self.last_mse = mse_avg
        if getattr(a, "defer_loss_sync", False):
            if dist_util.is_main_process() and self.pbar is not None:
                self.pbar.update(1)
            return total
        total_f = float(total.item())
        if dist_util.is_main_process() and self.pbar is not None:
            self.pbar.update(1)
            self.pbar.set_postfix(mse=f"{float(mse_avg.item()):.4f}")
        return total_f
