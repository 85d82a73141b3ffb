"""Data parallelism for the HIP denoisers: one process per GPU, gradient buckets all-reduced by RCCL over
xGMI on a side HIP stream while the hand-written backward is still producing the next bucket.

Stands where the reference wraps its model in torch DDP (main.py:347) and keeps DDP's surface: `.module`,
`forward`, `no_sync()`, 'module.'-prefixed state_dict.  torch DDP itself cannot be used: it hooks autograd's
per-parameter accumulation, and our denoisers run their whole backward inside one autograd node.

Buckets are contiguous ranges of the model's flat f32 gradient buffer (flat.py) in the order backward
finishes them: [head] -> [block L-1] ... [block 0] -> [embedders + all adaLN].  For DiT-B that is 14
all-reduces of 0.02..170 MB instead of DDP's ~21 x 25 MB, each launched the moment its last wgrad kernel
is enqueued.  xGMI is point-to-point (7 links/GPU): bucket size is chosen per model stage, not tuned for
NVSwitch; ring vs direct algorithm selection is left to RCCL.
"""
from contextlib import contextmanager

import torch
import torch.distributed as dist
import torch.nn as nn

from .flat import FlatModule


class DistributedDataParallel(nn.Module):
    def __init__(self, module, device_ids=None, output_device=None, process_group=None, broadcast=True):
        super().__init__()
        if not isinstance(module, FlatModule):
            raise TypeError("vaw_amd.DistributedDataParallel wraps FlatModule denoisers (e.g. vaw_amd.DiT)")
        if not dist.is_initialized():
            raise RuntimeError("init the process group first (vaw_amd.dist_util.setup_dist)")
        self.module = module
        self.pg = process_group
        self.world = dist.get_world_size(process_group)
        self._sync = True
        self._pending = []
        module.ensure_flat()
        self._cuda = module._flat.is_cuda
        self._comm = torch.cuda.Stream() if self._cuda else None
        self._backend_avg = dist.get_backend(process_group) == "nccl"     # RCCL has ReduceOp.AVG; gloo does not
        if broadcast:
            dist.broadcast(module._flat, src=0, group=process_group)     # one collective for all parameters
            if getattr(module, "_flat_shadow", None) is not None:
                module._shadow_version = None                            # the bf16 copy must follow the new weights
        self._ranges = self._stage_ranges()
        module.grad_ready_hook = self._on_stage

    # stage -> (start, end) element range of the flat gradient buffer that is final once `stage` fires
    def _stage_ranges(self):
        m = self.module
        bounds = getattr(m, "grad_stage_bounds", None)
        if bounds is None:
            return {0: (0, m._flat_n_train)}
        return bounds()

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    @contextmanager
    def no_sync(self):
        """Skip the all-reduce (gradient accumulation micro-steps), reference tools/trainer.py:94-101."""
        prev, self._sync = self._sync, False
        try:
            yield
        finally:
            self._sync = prev

    def _on_stage(self, stage):
        if not self._sync or self.world == 1:
            return
        rngs = self._ranges.get(stage)
        if rngs is not None and not isinstance(rngs, list):
            rngs = [rngs]
        for rng in rngs or ():
            if rng[1] <= rng[0]:
                continue
            g = self.module.flat_grads()[rng[0]:rng[1]]
            op = dist.ReduceOp.AVG if self._backend_avg else dist.ReduceOp.SUM
            if self._cuda:
                ev = torch.cuda.Event()
                ev.record()
                with torch.cuda.stream(self._comm):
                    self._comm.wait_event(ev)
                    w = dist.all_reduce(g, op=op, group=self.pg, async_op=True)
            else:
                w = dist.all_reduce(g, op=op, group=self.pg, async_op=True)
            self._pending.append((w, None if self._backend_avg else g))
        if stage == 0:
            self.finish()

    def finish(self):
        """Make the compute stream wait for every bucket (called at the end of backward)."""
        for w, g in self._pending:
            w.wait()
            if g is not None:
                g.div_(self.world)
        if self._cuda and self._pending:
            torch.cuda.current_stream().wait_stream(self._comm)
        self._pending = []
