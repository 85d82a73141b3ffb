"""Timestep importance samplers, same API as the reference's tools/resample.py (which nothing in the
reference calls: `Trainer` draws t through GaussianDiffusion.sample_t).  Host-side float64 numpy by nature:
T=1000 weights and a [T,10] loss history; the only device traffic is the batch of indices/weights going up
and, for the loss-aware sampler, one all_gather of (t, loss) pairs per step."""
from abc import ABC, abstractmethod

import numpy as np
import torch
import torch.distributed as dist


class ScheduleSampler(ABC):
    @abstractmethod
    def weights(self):
        """numpy array [T] of positive (unnormalised) weights."""

    def sample(self, batch_size, device):
        w = self.weights()
        p = w / np.sum(w)
        indices_np = np.random.choice(len(p), size=(batch_size,), p=p)
        indices = torch.from_numpy(indices_np).long().to(device)
        weights = torch.from_numpy(1 / (len(p) * p[indices_np])).float().to(device)
        return indices, weights


class UniformSampler(ScheduleSampler):
    def __init__(self, diffusion):
        self.diffusion = diffusion
        self._weights = np.ones([diffusion.num_timesteps])

    def weights(self):
        return self._weights


class LossAwareSampler(ScheduleSampler):
    def update_with_local_losses(self, local_ts, local_losses):
        """Every rank contributes its (t, loss) pairs; all ranks end with identical history.
        One padded all_gather of a [max_bs, 2] float64 tensor (the reference issues three collectives)."""
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            world = dist.get_world_size()
            n = torch.tensor([len(local_ts)], dtype=torch.int64, device=local_ts.device)
            sizes = [torch.zeros_like(n) for _ in range(world)]
            dist.all_gather(sizes, n)
            sizes = [int(s.item()) for s in sizes]
            mx = max(sizes)
            pack = torch.zeros(mx, 2, dtype=torch.float64, device=local_ts.device)
            pack[: len(local_ts), 0] = local_ts.double()
            pack[: len(local_ts), 1] = local_losses.double()
            out = [torch.zeros_like(pack) for _ in range(world)]
            dist.all_gather(out, pack)
            timesteps = [int(v) for o, s in zip(out, sizes) for v in o[:s, 0].tolist()]
            losses = [float(v) for o, s in zip(out, sizes) for v in o[:s, 1].tolist()]
        else:
            timesteps, losses = local_ts.tolist(), local_losses.tolist()
        self.update_with_all_losses(timesteps, losses)

    @abstractmethod
    def update_with_all_losses(self, ts, losses):
        ...


class LossSecondMomentResampler(LossAwareSampler):
    """w_t = sqrt(mean of the last `history_per_term` squared losses seen at t), mixed with a uniform floor once
    every timestep has a full history.  The history is a ring per timestep: the weight is a mean of squares,
    so it does not depend on the order in which the last 10 losses are stored."""

    def __init__(self, diffusion, history_per_term=10, uniform_prob=0.001):
        self.diffusion = diffusion
        self.history_per_term = history_per_term
        self.uniform_prob = uniform_prob
        T = diffusion.num_timesteps
        self._ring = np.zeros((T, history_per_term), dtype=np.float64)
        self._seen = np.zeros(T, dtype=np.int64)      # total losses ever recorded per timestep

    @property
    def _loss_counts(self):
        return np.minimum(self._seen, self.history_per_term)

    def _warmed_up(self):
        return bool((self._seen >= self.history_per_term).all())

    def weights(self):
        T = self.diffusion.num_timesteps
        if not self._warmed_up():
            return np.ones(T, dtype=np.float64)
        w = np.sqrt((self._ring ** 2).mean(axis=-1))
        w = w / w.sum() * (1 - self.uniform_prob)
        return w + self.uniform_prob / T

    def update_with_all_losses(self, ts, losses):
        H = self.history_per_term
        for t, loss in zip(ts, losses):
            self._ring[t, self._seen[t] % H] = loss
            self._seen[t] += 1


def create_named_schedule_sampler(name, diffusion):
    if name == "uniform":
        return UniformSampler(diffusion)
    if name == "loss-second-moment":
        return LossSecondMomentResampler(diffusion)
    raise NotImplementedError(f"unknown schedule sampler: {name}")
