"""Fused AdamW (+EMA, +global-norm clip, +bf16 shadow refresh) over a FlatModule's flat buffers.

Replaces, with ONE kernel launch per step (36 B/param + 2 B shadow):
  torch.optim.AdamW.step            (reference main.py:354; single-tensor update rule of torch 2.x)
  nn.utils.clip_grad_norm_          (reference tools/trainer.py:60-62) -- the norm is a second small kernel,
                                    the scale is applied inside the update, nothing syncs with the host
  ema(model, ema_model, decay)      (reference tools/trainer.py:12-18)
  optimizer.zero_grad()
It is a torch.optim.Optimizer, so LambdaLR (reference main.py:355) drives `param_groups[0]['lr']` as usual.
"""
import torch

from . import ops
from .flat import FlatModule


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        inner = getattr(model, "module", model)
        if not isinstance(inner, FlatModule):
            raise TypeError("FusedAdamW needs a vaw_amd FlatModule (e.g. vaw_amd.DiT); use torch.optim.AdamW otherwise")
        inner.ensure_flat()
        self.model = inner
        super().__init__(inner._flat_params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        n = inner._flat_n_train
        dev = inner._flat.device
        # ZeRO-1 (vaw_amd.DistributedDataParallel(..., shard_optimizer=True), wrapped BEFORE this optimizer is built): AdamW state
        # and the EMA exist only for this rank's chunk of every gradient bucket, stored back to back in compact buffers
        self.zero = getattr(inner, "_zero", None)
        if self.zero is not None:
            self._chunks = self.zero.owned_chunks()
            n = sum(hi - lo for lo, hi in self._chunks)
            self._ema_shard = self._ema_frozen = None
        self.exp_avg = torch.zeros(n, device=dev, dtype=torch.float32)
        self.exp_avg_sq = torch.zeros(n, device=dev, dtype=torch.float32)
        self._sumsq = torch.zeros(1, device=dev, dtype=torch.float32)
        self.step_count = 0
        self.ema_model, self.ema_decay = None, 0.0
        self.max_grad_norm = None          # set per step by the trainer (args.grad_clip)
        self._flat_ptr = inner._flat.data_ptr()
        # hipGraph mode (Trainer, args.hip_graph): lr and the bias corrections live in device memory, refreshed by
        # prepare_step() before every (captured or replayed) step; step() then neither counts nor reads host scalars
        self.device_hyper = None
        self._hyper_host = None

    def enable_device_hyper(self):
        dev = self.model._flat.device
        self.device_hyper = torch.zeros(4, device=dev, dtype=torch.float32)
        self._hyper_host = torch.zeros(4, dtype=torch.float32).pin_memory() if dev.type == "cuda" else torch.zeros(4)

    def prepare_step(self):
        """Host side of one step in device-hyper mode: count it and upload {lr, bc1, bc2}."""
        grp = self.param_groups[0]
        self.step_count += 1
        b1, b2 = grp["betas"]
        self._hyper_host[0] = grp["lr"]
        self._hyper_host[1] = 1.0 - b1 ** self.step_count
        self._hyper_host[2] = 1.0 - b2 ** self.step_count
        self.device_hyper.copy_(self._hyper_host, non_blocking=True)

    def attach_ema(self, ema_model, decay):
        """Fold `ema(model, ema_model, decay)` into the update kernel.  Layouts must match (deepcopy does)."""
        e = getattr(ema_model, "module", ema_model)
        if not isinstance(e, FlatModule):
            raise TypeError("attach_ema needs a FlatModule EMA copy")
        e.ensure_flat()
        if e._flat_offsets != self.model._flat_offsets:
            raise ValueError("EMA model layout differs from the trained model")
        self.ema_model, self.ema_decay = e, float(decay)

    def attach_ema_sharded(self, decay, ema_model=None):
        """ZeRO-1: every rank averages its own chunks (compact f32 shard, started from the EMA model's weights when this rank
        has one -- the reference keeps the EMA copy on rank 0 only, main.py:344 -- else from the trained weights);
        consolidate_ema() writes the gathered average into an EMA model."""
        assert self.zero is not None
        import torch.distributed as dist
        n = self.model._flat_n_train
        full = self.model._flat[:n].clone()
        if self.zero.rank == 0 and ema_model is not None:
            e = getattr(ema_model, "module", ema_model)
            e.ensure_flat()
            full.copy_(e._flat[:n])
        dist.broadcast(full, src=0, group=self.zero.pg)     # rank 0's copy is THE average (it may come from a checkpoint)
        self._ema_shard = torch.cat([full[lo:hi] for lo, hi in self._chunks]).clone()
        self.ema_decay = float(decay)
        # frozen entries (pos_embed) are averaged too by the reference's ema() (tools/trainer.py:12-18: every state_dict entry);
        # they are not sharded: the rank that holds an EMA model keeps doing that part itself
        self._ema_frozen = None
        if ema_model is not None:
            e = getattr(ema_model, "module", ema_model)
            if e._flat.numel() > n:
                self._ema_frozen = e

    def _step_sharded(self):
        """reduce-scattered gradients -> AdamW (+EMA) on this rank's chunks only -> all-gather of the updated weights."""
        m, z, grp = self.model, self.zero, self.param_groups[0]
        m.ensure_flat()
        if m._flat.data_ptr() != self._flat_ptr:
            raise RuntimeError("the model's flat buffer was rebuilt (moved device / deep-copied) after the optimizer was created")
        g = m.flat_grads()
        z.wait_gathers()
        if self.device_hyper is None:
            self.step_count += 1
        clip = self.max_grad_norm
        if clip:      # global norm: this rank's chunks hold 1 / world of the averaged gradient; the scalar is summed over ranks
            for i, (lo, hi) in enumerate(self._chunks):
                ops.sumsq(g[lo:hi], self._sumsq, accumulate=i > 0)
            import torch.distributed as dist
            dist.all_reduce(self._sumsq, group=z.pg)
        shadow = m._flat_shadow
        off = 0
        for lo, hi in self._chunks:
            k = hi - lo
            ops.adamw_ema_step(m._flat[lo:hi], g[lo:hi], self.exp_avg[off:off + k], self.exp_avg_sq[off:off + k],
                               None if self._ema_shard is None else self._ema_shard[off:off + k],
                               None if shadow is None else shadow[lo:hi], grp["lr"], grp["betas"][0], grp["betas"][1], grp["eps"],
                               grp["weight_decay"], self.step_count, self.ema_decay, self._sumsq if clip else None, clip, False,
                               hyper=self.device_hyper)
            off += k
        # the weights the kernels read: the bf16 shadow in throughput mode (2 B per parameter on the wire; the f32 masters of
        # other ranks' chunks go stale until consolidate()), the f32 parameters in parity mode
        if shadow is not None and not getattr(m, "_fp8", False):
            # (the small f32 all-reduce first: collectives of one group run in issue order, and the forward that follows waits
            # for this one at once but for the gathered buckets only as it reaches them)
            self._sync_f32_read_params()
            z.all_gather_chunks(shadow)
            self.master_stale = m._master_stale = True
        else:
            # f32 parity mode, and fp8 mode (whose e4m3 weight copies are quantised from the f32 masters)
            z.all_gather_chunks(m._flat)
            if shadow is not None:
                z.wait_gathers()                  # (the bf16 shadow is re-cast from the gathered masters by the next forward)
        if self._ema_shard is not None and self._ema_frozen is not None:
            n = m._flat_n_train
            ops.ema_update(self._ema_frozen._flat[n:], m._flat[n:], self.ema_decay)
        if shadow is None or not getattr(m, "_fp8", False):
            m.mark_shadow_fresh()
        else:
            m.mark_weights_changed()
        self.ema_done_in_step = self._ema_shard is not None

    def _sync_f32_read_params(self):
        """bf16 mode gathers only the bf16 shadow, but some parameters are read by the kernels as f32 straight from the master
        buffer: every bias / norm vector (GEMM and conv epilogues, GroupNorm) and the embedding tables (label rows added in f32).
        They are small (DiT-B: 0.95 M of 130 M elements): each rank contributes the elements it owns, zeros elsewhere, and one
        all-reduce(sum) hands everybody the exact f32 values."""
        import torch.distributed as dist
        m, z = self.model, self.zero
        hot = getattr(self, "_hot", None)
        if hot is None:
            emb = {n + ".weight" for n, mod in m.named_modules() if isinstance(mod, torch.nn.Embedding)}
            idx = [torch.arange(o, o + k) for n, p in zip(m._flat_names, m._flat_params) if (p.dim() == 1 or n in emb)
                   for o, k in [m._flat_offsets[n]]]
            idx = torch.cat(idx) if idx else torch.zeros(0, dtype=torch.long)
            own = torch.zeros(idx.numel(), dtype=torch.bool)
            for lo, hi in self._chunks:
                own |= (idx >= lo) & (idx < hi)
            dev = m._flat.device
            hot = self._hot = (idx.to(dev), own.to(dev))
        idx, own = hot
        if idx.numel() == 0:
            return
        vals = torch.where(own, m._flat[idx], torch.zeros((), device=idx.device))
        dist.all_reduce(vals, group=z.pg)
        m._flat[idx] = vals

    def consolidate(self):
        """ZeRO-1: bring every rank's f32 master parameters up to date (before state_dict / checkpoints / evaluation in f32)."""
        if self.zero is not None and getattr(self, "master_stale", False):
            self.zero.wait_gathers()
            self.zero.all_gather_chunks(self.model._flat, async_stream=False)
            self.master_stale = self.model._master_stale = False

    def consolidate_ema(self, ema_model):
        """ZeRO-1: gather the sharded average into `ema_model`'s trainable parameters (every rank that passes a model gets it)."""
        import torch.distributed as dist
        z = self.zero
        e = getattr(ema_model, "module", ema_model) if ema_model is not None else None
        full = torch.empty(self.model._flat_n_train, device=self.model._flat.device, dtype=torch.float32)
        off = 0
        for lo, hi in self._chunks:
            full[lo:hi] = self._ema_shard[off:off + hi - lo]
            off += hi - lo
        z.wait_gathers()
        z.all_gather_chunks(full, async_stream=False)
        if e is not None:
            e.ensure_flat()
            e._flat[: full.numel()].copy_(full)
            e.mark_weights_changed()

    def _gathered_moments(self):
        """Full-length exp_avg / exp_avg_sq (state_dict of a sharded optimizer)."""
        outs = []
        for shard in (self.exp_avg, self.exp_avg_sq):
            full = torch.zeros(self.model._flat_n_train, device=shard.device, dtype=torch.float32)
            off = 0
            for lo, hi in self._chunks:
                full[lo:hi] = shard[off:off + hi - lo]
                off += hi - lo
            self.zero.all_gather_chunks(full, async_stream=False)
            outs.append(full)
        return outs

    @torch.no_grad()
    def step(self, closure=None):
        assert closure is None
        if self.zero is not None:
            return self._step_sharded()
        m = self.model
        if getattr(m, "_zero", None) is not None:
            # the model reduce-SCATTERS its gradients (DistributedDataParallel(..., shard_optimizer=True)): only this rank's chunk of
            # every bucket is averaged.  An optimizer built before the wrap captured zero = None and would update every parameter
            # from mostly un-reduced local gradients, never gather, and let the ranks drift apart silently
            raise RuntimeError("this FusedAdamW was built BEFORE the model was wrapped with shard_optimizer=True: build the optimizer "
                               "after vaw_amd.DistributedDataParallel(model, shard_optimizer=True)")
        m.ensure_flat()
        if m._flat.data_ptr() != self._flat_ptr:
            raise RuntimeError("the model's flat buffer was rebuilt (moved device / deep-copied) after the optimizer was created")
        g = m.flat_grads()
        grp = self.param_groups[0]
        if self.device_hyper is None:
            self.step_count += 1
        n = m._flat_n_train
        clip = self.max_grad_norm
        if clip:
            ops.sumsq(g, self._sumsq)
        ema_flat = None
        if self.ema_model is not None:
            self.ema_model.ensure_flat()
            ema_flat = self.ema_model._flat
        shadow = m._flat_shadow
        ops.adamw_ema_step(m._flat[:n], g, self.exp_avg, self.exp_avg_sq, None if ema_flat is None else ema_flat[:n],
                           None if shadow is None else shadow[:n], grp["lr"], grp["betas"][0], grp["betas"][1], grp["eps"],
                           grp["weight_decay"], self.step_count, self.ema_decay, self._sumsq if clip else None, clip, False,
                           hyper=self.device_hyper)
        if ema_flat is not None and ema_flat.numel() > n:
            ops.ema_update(ema_flat[n:], m._flat[n:], self.ema_decay)   # frozen entries (pos_embed) are EMA'd too
        m.mark_shadow_fresh()
        if self.ema_model is not None:
            self.ema_model.mark_weights_changed()      # its bf16 / fp8 copies (periodic sampling, eval) follow the new average
        self.ema_done_in_step = ema_flat is not None

    def zero_grad(self, set_to_none=True):
        self.model.zero_grad_flat()

    # ---- checkpoint format: torch.optim.AdamW's, indexed like AdamW(model.parameters()) of the reference (main.py:354,
    # tools/utils.py:93-106), so optimizer states move between the reference and this engine in both directions ----
    def _ckpt_params(self):
        """[(name, param, trainable)] in model.parameters() order -- the index space of the reference's optimizer state."""
        m = self.model
        return [(n, p, n in m._flat_offsets and m._flat_offsets[n][0] < m._flat_n_train and p.requires_grad)
                for n, p in m.named_parameters()]

    def _moment_view(self, buf, name, p):
        m = self.model
        o, k = m._flat_offsets[name]
        return m._view_as_param(buf[o:o + k], p, m._flat_cl[name])

    def state_dict(self):
        self.model.ensure_flat()
        grp = {k: v for k, v in self.param_groups[0].items() if k != "params"}
        entries = self._ckpt_params()
        grp["params"] = list(range(len(entries)))
        state = {}
        if self.step_count > 0:
            # sharded: a collective -- every rank calls state_dict() and gets the full state
            exp_avg, exp_avg_sq = self._gathered_moments() if self.zero is not None else (self.exp_avg, self.exp_avg_sq)
            for i, (n, p, trainable) in enumerate(entries):
                if trainable:
                    state[i] = {"step": torch.tensor(float(self.step_count)),
                                "exp_avg": self._moment_view(exp_avg, n, p).contiguous().clone(),
                                "exp_avg_sq": self._moment_view(exp_avg_sq, n, p).contiguous().clone()}
        return {"state": state, "param_groups": [grp]}

    def load_state_dict(self, sd):
        self.model.ensure_flat()
        if self.zero is not None:      # load into full-length buffers, then keep this rank's chunks
            shards = (self.exp_avg, self.exp_avg_sq)
            n = self.model._flat_n_train
            self.exp_avg, self.exp_avg_sq = (torch.zeros(n, device=shards[0].device) for _ in range(2))
            try:
                self._load_state_dict_full(sd)
                full = (self.exp_avg, self.exp_avg_sq)
            finally:
                self.exp_avg, self.exp_avg_sq = shards
            for shard, f in zip(shards, full):
                shard.copy_(torch.cat([f[lo:hi] for lo, hi in self._chunks]))
            return
        self._load_state_dict_full(sd)

    def _load_state_dict_full(self, sd):
        flat = sd.get("vaw_flat")                      # round-1 private format
        if flat is not None:
            self.exp_avg.copy_(flat["exp_avg"])
            self.exp_avg_sq.copy_(flat["exp_avg_sq"])
            self.step_count = int(flat["step"])
        else:
            state = sd.get("state", {})
            entries = self._ckpt_params()
            groups = sd.get("param_groups", [])
            n_saved = sum(len(g.get("params", [])) for g in groups)
            if state and n_saved != len(entries):
                raise ValueError(f"optimizer state covers {n_saved} parameters, the model has {len(entries)}")
            steps = set()
            self.exp_avg.zero_()
            self.exp_avg_sq.zero_()
            for i, (n, p, trainable) in enumerate(entries):
                st = state.get(i, state.get(str(i)))
                if st is None:
                    continue
                if not trainable:
                    raise ValueError(f"optimizer state for frozen parameter {n}")
                if tuple(st["exp_avg"].shape) != tuple(p.shape):
                    raise ValueError(f"optimizer state of {n}: shape {tuple(st['exp_avg'].shape)} != {tuple(p.shape)}")
                self._moment_view(self.exp_avg, n, p).copy_(st["exp_avg"])
                self._moment_view(self.exp_avg_sq, n, p).copy_(st["exp_avg_sq"])
                steps.add(int(float(st["step"])))
            if len(steps) > 1:
                raise ValueError(f"per-parameter step counts differ ({sorted(steps)}): the fused kernel keeps one")
            self.step_count = steps.pop() if steps else 0
        for g_new, g_old in zip(self.param_groups, sd["param_groups"]):
            for k in ("lr", "betas", "eps", "weight_decay", "initial_lr"):
                if k in g_old:
                    g_new[k] = tuple(g_old[k]) if k == "betas" else g_old[k]
