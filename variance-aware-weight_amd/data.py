"""Input side of the training step: batches reach the GPU ahead of the step that uses them.

The reference moves each batch with `.to(device, non_blocking=True)` inside `Trainer._get_next_batch`
(tools/trainer.py:52-58) from an un-pinned DataLoader batch, i.e. a synchronous pageable copy on the compute stream.
`DevicePrefetcher` wraps any re-iterable loader of (images, labels): it stages batch k+1 into pinned host memory and
copies it on a side HIP stream while step k computes; `Trainer` then receives device tensors and its `.to()` is a no-op.
It keeps the loader surface `Trainer` relies on: re-iterable, `.sampler.set_epoch`.
"""
import math

import torch


class ShardedSampler:
    """Index stream of rank `rank` among `num_replicas` data-parallel ranks: the semantics of torch's DistributedSampler as
    the reference uses it (main.py:166-180: shuffle / drop_last flags, `set_epoch(step)` from Trainer.train_step :70-71).
    Per epoch: a permutation of range(n) from a generator seeded seed + epoch (or the identity), padded by wrapping around
    (or cut, with drop_last) to a multiple of num_replicas, of which this rank takes every num_replicas-th entry from `rank`."""

    def __init__(self, dataset_len, num_replicas, rank, shuffle=True, seed=0, drop_last=False):
        if not 0 <= rank < num_replicas:
            raise ValueError(f"rank {rank} outside [0, {num_replicas})")
        self.n, self.num_replicas, self.rank = int(dataset_len), int(num_replicas), int(rank)
        self.shuffle, self.seed, self.drop_last, self.epoch = shuffle, seed, drop_last, 0
        if drop_last and self.n % self.num_replicas:
            self.num_samples = math.ceil((self.n - self.num_replicas) / self.num_replicas)
        else:
            self.num_samples = math.ceil(self.n / self.num_replicas)
        self.total_size = self.num_samples * self.num_replicas

    def set_epoch(self, epoch):
        self.epoch = int(epoch)

    def __len__(self):
        return self.num_samples

    def __iter__(self):
        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(self.seed + self.epoch)
            order = torch.randperm(self.n, generator=g).tolist()
        else:
            order = list(range(self.n))
        if self.drop_last:
            order = order[: self.total_size]
        else:
            short = self.total_size - len(order)
            if short > 0:
                order += (order * math.ceil(short / max(len(order), 1)))[:short]
        return iter(order[self.rank: self.total_size: self.num_replicas])


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
