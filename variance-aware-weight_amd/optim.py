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

    @torch.no_grad()
    def step(self, closure=None):
        assert closure is None
        m = self.model
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
            for i, (n, p, trainable) in enumerate(entries):
                if trainable:
                    state[i] = {"step": torch.tensor(float(self.step_count)),
                                "exp_avg": self._moment_view(self.exp_avg, n, p).contiguous().clone(),
                                "exp_avg_sq": self._moment_view(self.exp_avg_sq, n, p).contiguous().clone()}
        return {"state": state, "param_groups": [grp]}

    def load_state_dict(self, sd):
        self.model.ensure_flat()
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
