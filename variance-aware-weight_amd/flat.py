"""Flat parameter storage for the HIP denoisers.

HBM layout: ONE contiguous f32 buffer holds every parameter of a model (each parameter is a view at a
256-byte-aligned offset), one f32 buffer of the same layout holds the gradients, and (throughput mode) one
bf16 buffer of the same layout holds the shadow weights the MFMA kernels read.  The fused AdamW+EMA kernel,
the gradient-norm kernel, the EMA kernel and the RCCL all-reduce buckets are then single passes over
contiguous ranges instead of ~150 per-tensor launches.  state_dict keys stay the reference's.
"""
import torch
import torch.nn as nn

from . import ops

_ALIGN = 64  # elements (256 B)


def _round_up(n, a=_ALIGN):
    return (n + a - 1) // a * a


class FlatModule(nn.Module):
    """nn.Module whose parameters live in one flat buffer.  Subclasses may override `_flat_groups()` to
    choose the order (a list of lists of parameter names; everything not named is appended)."""

    def _flat_groups(self):
        return []

    def _flat_channels_last(self, name, p):
        """True -> a 4-D parameter [Co,Ci,kh,kw] is STORED [Co][kh][kw][Ci] (torch channels_last) and exposed as a
        permuted view with the reference's shape."""
        return False

    @staticmethod
    def _view_as_param(buf, p, channels_last):
        if channels_last:
            co, ci, kh, kw = p.shape
            return buf.view(co, kh, kw, ci).permute(0, 3, 1, 2)
        return buf.view(p.shape)

    def _ordered_named_params(self):
        """-> (trainable, frozen) lists of (name, param, packed) where packed=True means "no alignment gap
        before this entry" (members of one group form a single contiguous matrix)."""
        named = dict(self.named_parameters())
        order, seen = [], set()
        for group in self._flat_groups():
            for i, n in enumerate(group):
                order.append((n, i > 0))
                seen.add(n)
        order += [(n, False) for n in named if n not in seen]
        train = [(n, named[n], pk) for n, pk in order if named[n].requires_grad]
        frozen = [(n, named[n], pk) for n, pk in order if not named[n].requires_grad]
        return train, frozen

    def _build_flat(self):
        train, frozen = self._ordered_named_params()
        dev = train[0][1].device
        offs, off = {}, 0
        for n, p, packed in train:
            if not packed:
                off = _round_up(off)
            offs[n] = (off, p.numel())
            off += p.numel()
        n_train = off = _round_up(off)
        for n, p, packed in frozen:
            off = _round_up(off)
            offs[n] = (off, p.numel())
            off += p.numel()
        off = _round_up(off)
        flat = torch.zeros(off, device=dev, dtype=torch.float32)
        self._flat_cl = {}
        with torch.no_grad():
            for n, p, _ in train + frozen:
                o, k = offs[n]
                if p.dtype != torch.float32:
                    raise TypeError(f"{n}: master parameters must be float32, got {p.dtype}")
                cl = self._flat_cl[n] = bool(p.dim() == 4 and self._flat_channels_last(n, p))
                v = self._view_as_param(flat[o:o + k], p, cl)
                v.copy_(p.data)
                p.data = v
        self._flat = flat
        self._flat_offsets = offs
        self._flat_n_train = n_train
        self._flat_grad = None
        self._flat_shadow = None
        self._shadow_version = None
        self._weights_epoch = 0
        self._flat_params = [p for _, p, _ in train]
        self._flat_names = [n for n, _, _ in train]
        self._flat_dirty = False

    def _apply(self, fn, recurse=True):
        r = super()._apply(fn, recurse)
        self._flat_dirty = True
        return r

    def ensure_flat(self):
        if getattr(self, "_flat_dirty", True) or getattr(self, "_flat", None) is None:
            self._build_flat()
            return
        # deepcopy / load paths can detach views: verify the first and last parameter still alias the buffer
        base = self._flat.data_ptr()
        for i in (0, -1):
            if self._flat_params[i].data_ptr() != base + 4 * self._flat_offsets[self._flat_names[i]][0]:
                self._build_flat()
                return

    def __deepcopy__(self, memo):
        # default deepcopy would copy every view separately; rebuild the flat buffer on the copy instead
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        import copy
        skip = {"_flat", "_flat_grad", "_flat_shadow", "_flat_params"}
        # per-batch workspaces / tapes / descriptors hold activations of the source model: the copy starts with none
        fresh = {"_ws": dict, "_ws_cur": lambda: None, "_adesc": lambda: None, "_tape": list, "_fp8_w": dict, "_fp8_epoch": lambda: None, "_fp8_wstates": lambda: None, "_fp8_wgroup": lambda: None, "_zero": lambda: None, "grad_ready_hook": lambda: None}
        for k, v in self.__dict__.items():
            new.__dict__[k] = None if k in skip else fresh[k]() if k in fresh else copy.deepcopy(v, memo)
        new._flat_dirty = True
        return new

    # ---- views -----------------------------------------------------------------------------
    def flat_params(self):
        self.ensure_flat()
        return self._flat

    def flat_train(self):
        self.ensure_flat()
        return self._flat[: self._flat_n_train]

    def flat_grads(self):
        self.ensure_flat()
        if self._flat_grad is None:
            self._flat_grad = torch.zeros(self._flat_n_train, device=self._flat.device, dtype=torch.float32)
        return self._flat_grad

    def grad_view(self, name):
        o, k = self._flat_offsets[name]
        return self.flat_grads()[o:o + k]

    def attach_grads(self):
        """Point every parameter's .grad at its slice of the flat gradient buffer."""
        g = self.flat_grads()
        for n, p in zip(self._flat_names, self._flat_params):
            if p.grad is None or p.grad.data_ptr() != g.data_ptr() + 4 * self._flat_offsets[n][0]:
                o, k = self._flat_offsets[n]
                p.grad = self._view_as_param(g[o:o + k], p, self._flat_cl[n])

    def grads_live(self):
        """True when gradients already hold a value that the next backward must add to
        (torch convention: .grad is None after optimizer.zero_grad())."""
        return self._flat_params[0].grad is not None

    def zero_grad_flat(self):
        """Drop gradients (set_to_none semantics): the next backward overwrites instead of accumulating."""
        for p in self._flat_params:
            p.grad = None

    # ---- sharded optimizer (ZeRO-1, bf16 mode): after a step only the bf16 shadow is gathered -- the f32 masters of OTHER ranks'
    # chunks are stale until optimizer.consolidate() / trainer.consolidate() (a collective: every rank calls it).  The flag lives
    # here, on the module, so that everything that would read or re-derive from the masters can refuse instead of silently
    # using the stale mix: state_dict(), a re-cast of the shadow, fp8 re-quantisation, set_compute_dtype.
    _master_stale = False

    def require_fresh_masters(self, what):
        if self._master_stale:
            raise RuntimeError(f"{what}: the f32 master parameters of other ranks' chunks are stale (sharded optimizer, bf16 mode) -- "
                               "call trainer.consolidate() / optimizer.consolidate() on EVERY rank first (it is a collective)")

    def state_dict(self, *args, **kwargs):
        self.require_fresh_masters("state_dict()")
        return super().state_dict(*args, **kwargs)

    def shadow_bf16(self):
        """bf16 copy of the flat parameters, refreshed when any parameter was modified through torch."""
        self.ensure_flat()
        ver = sum(p._version for p in self._flat_params)
        if self._flat_shadow is None:
            self._flat_shadow = torch.empty(self._flat.numel(), device=self._flat.device, dtype=torch.bfloat16)
            self._shadow_version = None
        if ver != self._shadow_version:
            self.require_fresh_masters("re-casting the bf16 shadow weights (a parameter was modified through torch)")
            ops.cast_bf16(self._flat, self._flat_shadow)
            self._shadow_version = ver
            self._weights_epoch += 1
        return self._flat_shadow

    def mark_weights_changed(self):
        """The flat buffer was rewritten through raw pointers (fused AdamW+EMA kernel on an EMA copy, `ops.ema_update`,
        a sharded optimizer's all-gather): parameter `_version`s did not move, so drop every derived copy explicitly --
        the bf16 shadow is re-cast and the fp8 weights (dit.py) re-quantised on the next forward."""
        self._shadow_version = None
        self._weights_epoch = getattr(self, "_weights_epoch", 0) + 1

    def mark_shadow_fresh(self):
        """Called by the fused optimizer after it rewrote parameters AND shadow in one pass."""
        self._weights_epoch += 1          # (other derived copies -- the fp8 weights of dit.py -- follow this counter)
        if self._flat_shadow is not None:
            self._shadow_version = sum(p._version for p in self._flat_params)
