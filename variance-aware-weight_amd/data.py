"""Input side of the training step: batches reach the GPU ahead of the step that uses them.

The reference moves each batch with `.to(device, non_blocking=True)` inside `Trainer._get_next_batch`
(tools/trainer.py:52-58) from an un-pinned DataLoader batch, i.e. a synchronous pageable copy on the compute stream.
`DevicePrefetcher` wraps any re-iterable loader of (images, labels): it stages batch k+1 into pinned host memory and
copies it on a side HIP stream while step k computes; `Trainer` then receives device tensors and its `.to()` is a no-op.
It keeps the loader surface `Trainer` relies on: re-iterable, `.sampler.set_epoch`.
"""
import torch


class DevicePrefetcher:
    def __init__(self, loader, device, depth=2):
        self.loader = loader
        self.device = torch.device(device)
        self.depth = max(1, int(depth))
        self._cuda = self.device.type == "cuda"
        self._stream = torch.cuda.Stream(self.device) if self._cuda else None

    @property
    def sampler(self):
        return getattr(self.loader, "sampler", None)

    def __len__(self):
        return len(self.loader)

    def _stage(self, batch):
        """Start the host-to-device copy of one batch on the side stream; returns (device tensors, event)."""
        if not self._cuda:
            return tuple(batch), None
        out = []
        with torch.cuda.stream(self._stream):
            for t in batch:
                if torch.is_tensor(t) and not t.is_cuda:
                    t = t.pin_memory() if not t.is_pinned() else t
                    t = t.to(self.device, non_blocking=True)
                out.append(t)
            ev = torch.cuda.Event()
            ev.record(self._stream)
        return tuple(out), ev

    def __iter__(self):
        it = iter(self.loader)
        queue = []
        try:
            while len(queue) < self.depth:
                queue.append(self._stage(next(it)))
        except StopIteration:
            it = None
        while queue:
            batch, ev = queue.pop(0)
            if it is not None:
                try:
                    queue.append(self._stage(next(it)))
                except StopIteration:
                    it = None
            if ev is not None:
                torch.cuda.current_stream(self.device).wait_event(ev)      # the step's stream waits; the host does not
                for t in batch:
                    if torch.is_tensor(t):
                        t.record_stream(torch.cuda.current_stream(self.device))
            yield batch
