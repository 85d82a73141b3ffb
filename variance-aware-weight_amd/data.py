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


class LatentH5Dataset:
    """The latent dataset of the reference's ImageNet-256 runs: datasets/data_loader.py:62-81 (`Latent`), written by
    preprocessing/encode_latent.py:95-126 --

        '<split>_latents'  float32 [N, 2 C, H, W]   cat[posterior mean, posterior std] of the SD-VAE (8 x 32 x 32 for 256-px images)
        '<split>_labels'   uint16  [N]

    `__getitem__(i)` -> (latent f32 tensor, label int64 tensor), exactly what `Latent.__getitem__` hands the DataLoader (the mean/std
    pair is sampled and scaled later, by `sample_from_latent` in `Trainer.train_step`).  `source` is a path (opened with h5py, which
    is imported only then -- it is not part of this image) or ANY mapping-like handle with those two keys whose values support
    `len()` and integer / slice / sorted-index-array `__getitem__` (an open `h5py.File`, a dict of numpy arrays, a zarr group ...).
    Unlike the reference, which re-opens the file for every sample, the handle is opened once per process (lazily, so the object
    can be pickled into DataLoader workers before any file is open) and `batch(indices)` reads a whole batch with ONE sorted
    fancy-index read per array (h5py needs increasing indices; the rows are put back in request order).
    Real-file parity is unpinned here (no h5py, no file): the tests pin the contract above on an in-memory stand-in."""

    def __init__(self, source, dataset_type="train"):
        self.source, self.dataset_type = source, dataset_type
        self._h = None
        self._lat_key, self._lab_key = f"{dataset_type}_latents", f"{dataset_type}_labels"
        self.num_samples = len(self._handle()[self._lat_key])
        if len(self._handle()[self._lab_key]) != self.num_samples:
            raise ValueError(f"{self._lat_key} and {self._lab_key} differ in length")
        if isinstance(source, (str, bytes)) or hasattr(source, "__fspath__"):
            self._close()                      # re-opened lazily in whichever process reads (DataLoader workers)

    def _handle(self):
        if self._h is None:
            src = self.source
            if isinstance(src, (str, bytes)) or hasattr(src, "__fspath__"):
                try:
                    import h5py
                except ImportError as e:       # pragma: no cover - h5py is absent from the build image
                    raise ImportError("LatentH5Dataset(path) needs h5py; pass an open mapping-like handle instead") from e
                src = h5py.File(src, "r")
            self._h = src
        return self._h

    def _close(self):
        h, self._h = self._h, None
        if h is not None and h is not self.source and hasattr(h, "close"):
            h.close()

    def __getstate__(self):
        d = dict(self.__dict__)
        if d["_h"] is not d["source"]:
            d["_h"] = None                     # an open file does not travel; the worker opens its own
        return d

    def __len__(self):
        return self.num_samples

    def __getitem__(self, idx):
        import numpy as np
        h = self._handle()
        latent = np.asarray(h[self._lat_key][idx], dtype=np.float32)
        label = np.asarray(h[self._lab_key][idx])
        return torch.tensor(latent.copy(), dtype=torch.float32), torch.tensor(label.astype(np.int64), dtype=torch.long)

    def batch(self, indices):
        """(latents [B, 2C, H, W] f32, labels [B] int64) for `indices` in the order given: one read per array."""
        import numpy as np
        idx = np.asarray(list(indices), dtype=np.int64)
        order = np.argsort(idx, kind="stable")
        uniq, inverse = np.unique(idx[order], return_inverse=True)      # h5py: strictly increasing, no repeats
        h = self._handle()
        lat = np.asarray(h[self._lat_key][uniq], dtype=np.float32)[inverse]
        lab = np.asarray(h[self._lab_key][uniq])[inverse]
        back = np.empty_like(order)
        back[order] = np.arange(len(order))
        return torch.from_numpy(np.ascontiguousarray(lat[back])), torch.from_numpy(lab[back].astype(np.int64))


class LatentBatchLoader:
    """Re-iterable loader of (latents, labels) batches over a LatentH5Dataset, driven by a sampler's index stream (a
    `ShardedSampler` in data-parallel runs: main.py:173-180's DistributedSampler + DataLoader(batch_size // world_size,
    drop_last=True)).  One `dataset.batch()` read per step; wrap it in `DevicePrefetcher` to overlap read and copy with the step.
    Keeps the surface `Trainer` uses: `.sampler.set_epoch`, `__iter__`, `__len__`."""

    def __init__(self, dataset, batch_size, sampler=None, drop_last=True):
        self.dataset, self.batch_size, self.drop_last = dataset, int(batch_size), drop_last
        self.sampler = sampler if sampler is not None else ShardedSampler(len(dataset), 1, 0, shuffle=False)

    def __len__(self):
        n = len(self.sampler)
        return n // self.batch_size if self.drop_last else math.ceil(n / self.batch_size)

    def __iter__(self):
        buf = []
        for i in self.sampler:
            buf.append(i)
            if len(buf) == self.batch_size:
                yield self.dataset.batch(buf)
                buf = []
        if buf and not self.drop_last:
            yield self.dataset.batch(buf)
