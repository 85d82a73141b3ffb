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
    """bucket_dtype: "f32" (reference DDP semantics: f32 buckets, main.py:347-348) or "bf16": each bucket is cast to bf16 on
    the side stream, all-reduced at half the wire bytes (xGMI is per-link bound: SURVEY.md §5) and widened back into the f32
    gradient buffer before the optimizer reads it.  The sum itself is then taken in bf16 by RCCL: a throughput option, off
    in parity runs."""

    def __init__(self, module, device_ids=None, output_device=None, process_group=None, broadcast=True, bucket_dtype="f32",
                 shard_optimizer=False, collectives="torch"):
        super().__init__()
        if not isinstance(module, FlatModule):
            raise TypeError("vaw_amd.DistributedDataParallel wraps FlatModule denoisers (e.g. vaw_amd.DiT)")
        if not dist.is_initialized():
            raise RuntimeError("init the process group first (vaw_amd.dist_util.setup_dist)")
        if bucket_dtype not in ("f32", "bf16"):
            raise ValueError("bucket_dtype must be 'f32' or 'bf16'")
        if collectives not in ("torch", "direct"):
            raise ValueError("collectives must be 'torch' (torch.distributed) or 'direct' (the library's own RCCL communicator)")
        if collectives == "direct" and shard_optimizer:
            raise ValueError("collectives='direct' covers the bucket all-reduce; the sharded optimizer's collectives go through torch.distributed")
        self.module = module
        self.pg = process_group
        self.world = dist.get_world_size(process_group)
        self._sync = True
        self._pending = []
        module.ensure_flat()
        self._cuda = module._flat.is_cuda
        self._comm = torch.cuda.Stream() if self._cuda else None
        self._backend_avg = dist.get_backend(process_group) == "nccl"     # RCCL has ReduceOp.AVG; gloo does not
        # collectives="direct": the bucket all-reduce is vaw_allreduce_bucket_start / _wait (csrc/collective.hip: the library's own
        # RCCL communicator and side stream, no torch.distributed on the data path); the process group only carries the 128-byte
        # communicator id at start-up.  Off by default: without a multi-GPU node it has run at world size 1 only.
        self._direct = collectives == "direct"
        if self._direct:
            if not (self._cuda and self._backend_avg):
                raise ValueError("collectives='direct' needs the denoiser on a GPU and an RCCL ('nccl') process group")
            self._init_direct()
        self._wire = None
        if bucket_dtype == "bf16":
            self._wire = torch.empty(module._flat_n_train, device=module._flat.device, dtype=torch.bfloat16)
        if broadcast:
            dist.broadcast(module._flat, src=0, group=process_group)     # one collective for all parameters
            if getattr(module, "_flat_shadow", None) is not None:
                module._shadow_version = None                            # the bf16 copy must follow the new weights
        self._ranges = self._stage_ranges()
        # shard_optimizer (ZeRO-1; SURVEY.md §8(e), beyond the reference, off by default): every gradient bucket is REDUCE-SCATTERED
        # instead of all-reduced -- rank r ends up with the averaged gradients of the r-th of `world` equal chunks of each bucket
        # --, vaw_amd.FusedAdamW then updates only those chunks (AdamW state and EMA for 1 / world of the parameters per rank) and
        # all-gathers the updated weights bucket by bucket: half the wire bytes of all-reduce + the optimizer's HBM pass divided
        # by `world` (at 32 images per GPU the un-sharded AdamW + EMA pass is 13-15 % of the DiT-B/4 step)
        self.shard_optimizer = bool(shard_optimizer) and self.world > 1
        self.rank = dist.get_rank(process_group)
        if self.shard_optimizer:
            for lo, hi in self.bucket_ranges():
                if (hi - lo) % (8 * self.world):
                    raise ValueError(f"shard_optimizer: bucket [{lo}, {hi}) does not split into {self.world} chunks of whole 32-byte groups")
            object.__setattr__(module, "_zero", self)       # (not a child module: the wrapper contains the module, not the reverse)
        n_buckets = sum(len(r) if isinstance(r, list) else 1 for r in self._ranges.values())
        self._events = [torch.cuda.Event() for _ in range(n_buckets)] if self._cuda else []   # one per bucket, reused every step
        self._issued = 0
        # While RCCL's kernels hold CUs of their own, the persistent GEMMs leave that many alone (gemm_p8.hip: p8_num_cus).  64 covers
        # RCCL's largest channel count; its real use on an 8-GPU xGMI node could not be measured on the one-GPU boxes, and with the
        # stand-in of tools/contention_bench.py leaving out 64 costs the DiT-B/4 step no more than leaving out 32 or 16 (the 768-item
        # launches need four rounds either way).  VAW_DDP_RESERVE_CUS overrides (0 = off).  Only with the RCCL backend: gloo reduces
        # on the host.
        import os
        explicit = "VAW_DDP_RESERVE_CUS" in os.environ        # (set explicitly it also applies to a gloo rehearsal: tests)
        self._reserve = int(os.environ.get("VAW_DDP_RESERVE_CUS", "64")) if (self._cuda and (self._backend_avg or explicit)) else 0
        self._reserved_now = False            # ... for the reduce-scatters / all-reduces of a backward in flight
        self._reserved_gather = False         # ... for the all-gathers of the sharded optimizer that overlap the next forward
        self._gather_events, self._gather_waiting = {}, []
        module.grad_ready_hook = self._on_stage

    # stage -> (start, end) element range of the flat gradient buffer that is final once `stage` fires
    def _stage_ranges(self):
        m = self.module
        bounds = getattr(m, "grad_stage_bounds", None)
        if bounds is None:
            return {0: (0, m._flat_n_train)}
        return bounds()

    def bucket_ranges(self):
        """Every bucket [lo, hi) of the flat gradient buffer, ascending (a partition of [0, n_train))."""
        out = []
        for r in self._ranges.values():
            out += [tuple(x) for x in (r if isinstance(r, list) else [r]) if x[1] > x[0]]
        return sorted(out)

    def owned_chunks(self, rank=None):
        """shard_optimizer: the element ranges of the flat buffers this rank keeps optimizer state for (one chunk per bucket)."""
        r = self.rank if rank is None else rank
        return [(lo + r * ((hi - lo) // self.world), lo + (r + 1) * ((hi - lo) // self.world)) for lo, hi in self.bucket_ranges()]

    def forward(self, *args, **kwargs):
        if self._reserved_now or self._pending or self._issued:
            self._abandon()          # a backward that raised never reached finish(): do not run the next step on its leftovers
        if self._gather_waiting and not getattr(self.module, "stagewise_weight_waits", False):
            self.wait_gathers()      # the module does not ask for its weights stage by stage (wait_stage): all of them, now
        return self.module(*args, **kwargs)

    def _apply_reservation(self):
        from . import _lib
        _lib.lib().vaw_p8_set_reserved_cus(self._reserve if (self._reserved_now or self._reserved_gather) else 0)

    @property
    def reserved_cus(self):
        """CUs the persistent GEMM grids leave to the collectives during a synchronised backward (0 = none)."""
        return self._reserve

    def _abandon(self):
        """Drop the bookkeeping of an interrupted backward: wait for its collectives, give the reserved CUs back."""
        for work, _, _ in self._pending:
            try:
                work.wait()
            except Exception:
                pass
        self._pending, self._issued = [], 0
        if self._reserved_now:
            self._reserved_now = False
            self._apply_reservation()

    @contextmanager
    def no_sync(self):
        """Skip the all-reduce (gradient accumulation micro-steps), reference tools/trainer.py:94-101."""
        prev, self._sync = self._sync, False
        try:
            yield
        finally:
            self._sync = prev

    def _to_wire(self, g, lo, hi):
        """f32 gradient range -> its bf16 wire buffer (current stream)."""
        w = self._wire[lo:hi]
        if self._cuda:
            from . import ops
            ops.cast_bf16(g, w)
        else:
            w.copy_(g)
        return w

    def _from_wire(self, w, g):
        """Reduced bf16 bucket -> f32 gradients, with the 1/world of backends that can only sum (current stream)."""
        scale = 1.0 if self._backend_avg else 1.0 / self.world
        if self._cuda:
            from . import ops
            ops.uncast_bf16(w, g, scale)
        else:
            g.copy_(w.float() * scale)

    def _init_direct(self):
        """Every rank joins the library's RCCL communicator; rank 0's id travels over the process group."""
        import ctypes

        from . import _lib as L
        lib = L.lib()
        if lib.vaw_comm_world() == self.world:
            return                                              # (one communicator per process: a second wrapper shares it)
        ident = torch.zeros(128, dtype=torch.uint8)
        if self.rank_of_group() == 0:
            raw = (ctypes.c_ubyte * 128)()
            L.check(lib.vaw_comm_unique_id(raw), "comm_unique_id")
            ident = torch.tensor(list(raw), dtype=torch.uint8)
        ident = ident.to(self.module._flat.device)
        dist.broadcast(ident, src=0, group=self.pg)
        raw = (ctypes.c_ubyte * 128)(*ident.cpu().tolist())
        if lib.vaw_comm_world():
            L.check(lib.vaw_comm_destroy(), "comm_destroy")
        L.check(lib.vaw_comm_init(raw, self.rank_of_group(), self.world), "comm_init")

    def rank_of_group(self):
        return dist.get_rank(self.pg)

    class _DirectWork:
        """What `_retire` waits on in direct mode: the current stream joins the library's collective stream."""

        def wait(self):
            from . import _lib as L
            L.check(L.lib().vaw_allreduce_bucket_wait(L.stream_ptr()), "allreduce_bucket_wait")

    def _reduce(self, g, buf, op):
        """The bucket's collective.  -> (work, the f32 gradient range the result belongs to, the buffer it arrives in).
        all-reduce: the whole bucket; shard_optimizer: reduce-scatter, this rank's chunk of the bucket only."""
        if self._direct:
            from . import _lib as L
            dt = L.BF16 if buf.dtype == torch.bfloat16 else L.F32
            L.check(L.lib().vaw_allreduce_bucket_start(L.ptr(buf), buf.numel(), dt, L.stream_ptr()), "allreduce_bucket_start")
            return self._DirectWork(), g, buf
        if not self.shard_optimizer:
            return dist.all_reduce(buf, op=op, group=self.pg, async_op=True), g, buf
        c = g.numel() // self.world
        g_out = g[self.rank * c:(self.rank + 1) * c]
        if self._backend_avg:          # RCCL: in place (the output is this rank's slot of the input)
            out = g_out if buf is g else buf[self.rank * c:(self.rank + 1) * c]
        else:                          # gloo: separate output
            out = torch.empty(c, device=buf.device, dtype=buf.dtype)
        work = dist.reduce_scatter_tensor(out, buf, op=op, group=self.pg, async_op=True)
        return work, g_out, out

    def _retire(self, entry):
        """Wait for one bucket's collective and put its result where the optimizer reads it (current stream)."""
        work, g, w = entry
        work.wait()
        if w is not None and w.dtype == torch.bfloat16:
            self._from_wire(w, g)
        elif w is not None:                      # f32 result that arrived beside the gradient buffer (gloo reduce-scatter)
            g.copy_(w if self._backend_avg else w / self.world)
        elif not self._backend_avg:
            g.div_(self.world)

    def _on_stage(self, stage):
        if not self._sync or self.world == 1:
            return
        rngs = self._ranges.get(stage)
        if rngs is not None and not isinstance(rngs, list):
            rngs = [rngs]
        op = dist.ReduceOp.AVG if self._backend_avg else dist.ReduceOp.SUM
        for lo, hi in rngs or ():
            if hi <= lo:
                continue
            g = self.module.flat_grads()[lo:hi]
            if self._reserve and not self._reserved_now:
                self._reserved_now = True
                self._apply_reservation()
            if self._cuda:
                ev = self._events[self._issued % len(self._events)]
                self._issued += 1
                ev.record()
                with torch.cuda.stream(self._comm):
                    # RCCL work.wait() only orders the side stream behind the collective: retiring the previous bucket here
                    # widens it back to f32 while backward is still running (gloo would block the host: it retires in finish)
                    if self._backend_avg and self._pending:
                        self._retire(self._pending.pop(0))
                    self._comm.wait_event(ev)
                    buf = self._to_wire(g, lo, hi) if self._wire is not None else g
                    work, g_out, buf = self._reduce(g, buf, op)
            else:
                buf = self._to_wire(g, lo, hi) if self._wire is not None else g
                work, g_out, buf = self._reduce(g, buf, op)
            self._pending.append((work, g_out, buf if (self._wire is not None or buf is not g_out) else None))
        if stage == 0:
            self.finish()

    def gather_stages(self):
        """[(stage, [bucket ranges])] in the order a FORWARD pass first reads the parameters: stage 0 (embedders; DiT: + the
        packed adaLN matrix, whose late rows are stage "ada_hi") -> the stages in ascending order (DiT: block l is stage l + 1,
        the head depth + 1; UNet: middle_block 2, decoder 3)."""
        keys = sorted((k for k in self._ranges if isinstance(k, int)))
        named = [k for k in self._ranges if not isinstance(k, int)]
        order = keys[:1] + named + keys[1:] if keys and keys[0] == 0 else named + keys
        out = []
        for k in order:
            r = self._ranges[k]
            rs = [tuple(x) for x in (r if isinstance(r, list) else [r]) if x[1] > x[0]]
            if rs:
                out.append((k, sorted(rs)))
        return out

    def all_gather_chunks(self, flat, async_stream=True):
        """shard_optimizer: every rank has rewritten its chunk of each bucket of `flat` (a full-length flat buffer: bf16 shadow
        or f32 parameters); all-gather the buckets so that every rank holds all of it again.  On the GPU the collectives run on
        the side stream behind the current one, stage by stage in the order the next forward reads the weights, each stage with
        its own event: a module that sets `stagewise_weight_waits` calls wait_stage(k) right before its first read of stage k
        (vaw_amd.DiT: only the embedder / adaLN stage is waited for at the start of the step, the blocks' weights arrive under
        the forward of the blocks before them); everybody else gets wait_gathers() from forward().  Synchronous otherwise."""
        def gather(record):
            for k, rngs in self.gather_stages():
                for lo, hi in rngs:
                    c = (hi - lo) // self.world
                    src = flat[lo + self.rank * c: lo + (self.rank + 1) * c]
                    if not self._backend_avg:
                        src = src.clone()                 # gloo: no in-place form
                    dist.all_gather_into_tensor(flat[lo:hi], src, group=self.pg)
                if record:
                    ev = self._gather_events.get(k)
                    if ev is None:
                        ev = self._gather_events[k] = torch.cuda.Event()
                    ev.record()
                    self._gather_waiting.append(k)
        if self._cuda and async_stream:
            self.wait_gathers()                           # (a second gather before anybody read the first: keep the events simple)
            self._comm.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._comm):
                gather(True)
            if self._reserve and not self._reserved_gather:
                self._reserved_gather = True              # RCCL's kernels run beside the forward now: leave them their CUs
                self._apply_reservation()
        else:
            gather(False)

    def wait_stage(self, stage):
        """The current stream waits until the gathered weights of `stage` (and of every stage gathered before it) have arrived."""
        if stage in self._gather_waiting:
            torch.cuda.current_stream().wait_event(self._gather_events[stage])
            del self._gather_waiting[: self._gather_waiting.index(stage) + 1]
            if not self._gather_waiting:
                self._gathers_done()

    def wait_gathers(self):
        if self._gather_waiting:
            torch.cuda.current_stream().wait_stream(self._comm)
            self._gather_waiting = []
            self._gathers_done()

    def _gathers_done(self):
        if self._reserved_gather:
            self._reserved_gather = False
            self._apply_reservation()

    def finish(self):
        """Retire every outstanding bucket and make the compute stream wait for the side stream (end of backward)."""
        if self._cuda:
            with torch.cuda.stream(self._comm):
                for entry in self._pending:
                    self._retire(entry)
            if self._pending or self._issued:
                torch.cuda.current_stream().wait_stream(self._comm)
        else:
            for entry in self._pending:
                self._retire(entry)
        self._pending = []
        self._issued = 0
        if self._reserved_now:
            self._reserved_now = False
            self._apply_reservation()
