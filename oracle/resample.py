"""Oracle (test infrastructure): restatement of the timestep importance samplers.

Follows /root/reference/tools/resample.py: create_named_schedule_sampler :9-21,
ScheduleSampler.sample :43-59, UniformSampler :62-68, LossAwareSampler :71-129
(single-process branch), LossSecondMomentResampler :132-162.  Host-side float64
numpy in the reference as well; nothing here reaches a GPU.  Pinned by
tests/golden/resample.json.
"""
import numpy as np
import torch


class ScheduleSampler:
    def weights(self):
        raise NotImplementedError

    def sample(self, batch_size, device):
        w = self.weights()
        p = w / np.sum(w)
        idx = np.random.choice(len(p), size=(batch_size,), p=p)
        return (torch.from_numpy(idx).long().to(device),
                torch.from_numpy(1 / (len(p) * p[idx])).float().to(device))


class UniformSampler(ScheduleSampler):
    def __init__(self, diffusion):
        self.diffusion = diffusion
        self._weights = np.ones([diffusion.num_timesteps])

    def weights(self):
        return self._weights


class LossSecondMomentResampler(ScheduleSampler):
    def __init__(self, diffusion, history_per_term=10, uniform_prob=0.001):
        self.diffusion = diffusion
        self.history_per_term = history_per_term
        self.uniform_prob = uniform_prob
        self._loss_history = np.zeros([diffusion.num_timesteps, history_per_term], dtype=np.float64)
        self._loss_counts = np.zeros([diffusion.num_timesteps], dtype=int)

    def _warmed_up(self):
        return (self._loss_counts == self.history_per_term).all()

    def weights(self):
        if not self._warmed_up():
            return np.ones([self.diffusion.num_timesteps], dtype=np.float64)
        w = np.sqrt(np.mean(self._loss_history ** 2, axis=-1))
        w /= np.sum(w)
        w *= 1 - self.uniform_prob
        w += self.uniform_prob / len(w)
        return w

    def update_with_local_losses(self, local_ts, local_losses):
        self.update_with_all_losses(local_ts.tolist(), local_losses.tolist())

    def update_with_all_losses(self, ts, losses):
        for t, l in zip(ts, losses):
            n = self._loss_counts[t]
            if n == self.history_per_term:
                self._loss_history[t, :-1] = self._loss_history[t, 1:]
                self._loss_history[t, -1] = l
            else:
                self._loss_history[t, n] = l
                self._loss_counts[t] = n + 1


def create_named_schedule_sampler(name, diffusion):
    if name == "uniform":
        return UniformSampler(diffusion)
    if name == "loss-second-moment":
        return LossSecondMomentResampler(diffusion)
    raise NotImplementedError(f"unknown schedule sampler: {name}")
