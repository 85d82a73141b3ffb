"""Timestep respacing for the sampling side: same names and behaviour as the reference's tools/respace.py
(space_timesteps :8-62, SpacedDiffusion :65-112, _WrappedModel :115-130) over the HIP-backed GaussianDiffusion."""
import numpy as np
import torch

from .gaussian_diffusion import GaussianDiffusion


def _ddim_stride(num_timesteps, count):
    """Smallest integer stride whose strided range over [0, T) has exactly `count` entries (the DDIM paper's spacing)."""
    stride = -(-num_timesteps // count) if count > 0 else 0          # ceil(T / count): the only candidate that can be smallest
    if not (1 <= stride < num_timesteps) or -(-num_timesteps // stride) != count:
        raise ValueError(f"cannot create exactly {num_timesteps} steps with an integer stride")
    return stride


def space_timesteps(num_timesteps, section_counts):
    """Timesteps of the base process to keep (reference tools/respace.py:8-62).

    "ddimN" keeps every s-th step for the smallest s that yields N steps.  Otherwise [0, T) is cut into len(counts)
    sections as equal as possible (the first T % len get one more) and section i keeps counts[i] steps spread from its
    first to its last index: offsets are the running float64 sums of (size - 1) / (count - 1), rounded half-to-even."""
    if isinstance(section_counts, str):
        if section_counts.startswith("ddim"):
            return set(range(0, num_timesteps, _ddim_stride(num_timesteps, int(section_counts[4:]))))
        section_counts = [int(tok) for tok in section_counts.split(",")]
    counts = np.asarray(section_counts, dtype=np.int64)
    sizes = np.full(len(counts), num_timesteps // len(counts), dtype=np.int64)
    sizes[: num_timesteps % len(counts)] += 1
    short = np.nonzero(sizes < counts)[0]
    if short.size:
        raise ValueError(f"cannot divide section of {int(sizes[short[0]])} steps into {int(counts[short[0]])}")
    firsts = np.cumsum(sizes) - sizes
    kept = set()
    for first, size, count in zip(firsts.tolist(), sizes.tolist(), counts.tolist()):
        if count <= 0:
            continue
        step = 1.0 if count == 1 else (size - 1) / (count - 1)
        offsets = np.concatenate(([0.0], np.cumsum(np.full(count - 1, step, dtype=np.float64))))   # sequential sums
        kept.update((first + np.rint(offsets).astype(np.int64)).tolist())
    return kept


class SpacedDiffusion(GaussianDiffusion):
    """The base process restricted to `use_timesteps` (reference tools/respace.py:65-112): the kept cumulative alphas
    define new betas  beta'_j = 1 - abar[k_j] / abar[k_{j-1}]  (abar[k_{-1}] = 1), and models are wrapped so that they
    still receive the ORIGINAL timestep values k_j."""

    def __init__(self, use_timesteps, **kwargs):
        base_betas = np.asarray(kwargs["betas"], dtype=np.float64)
        self.original_num_steps = int(base_betas.shape[0])
        self.use_timesteps = set(use_timesteps)
        keep = np.array(sorted(k for k in self.use_timesteps if 0 <= k < self.original_num_steps), dtype=np.int64)
        self.timestep_map = keep.tolist()
        abar = np.cumprod(1.0 - base_betas, axis=0)[keep]
        kwargs["betas"] = 1 - abar / np.concatenate(([1.0], abar[:-1]))
        super().__init__(**kwargs)

    def _reverse_step(self, kind, model, *args, **kwargs):       # p_mean_variance / p_sample / ddim_sample all come here
        return super()._reverse_step(kind, self._wrap_model(model), *args, **kwargs)

    def training_losses(self, model, *args, **kwargs):
        return super().training_losses(self._wrap_model(model), *args, **kwargs)

    def _wrap_model(self, model):
        if isinstance(model, _WrappedModel):
            return model
        return _WrappedModel(model, self.timestep_map, self.rescale_timesteps, self.original_num_steps)

    def _scale_timesteps(self, t):
        return t            # scaling is done by the wrapped model


class _WrappedModel:
    def __init__(self, model, timestep_map, rescale_timesteps, original_num_steps):
        self.model = model
        self.timestep_map = timestep_map
        self.rescale_timesteps = rescale_timesteps
        self.original_num_steps = original_num_steps
        self._map = {}

    def parameters(self):
        return self.model.parameters()

    def __call__(self, x, ts, **kwargs):
        key = (str(ts.device), ts.dtype)
        m = self._map.get(key)
        if m is None:
            m = self._map[key] = torch.tensor(self.timestep_map, device=ts.device, dtype=ts.dtype)
        new_ts = m[ts]
        if self.rescale_timesteps:
            new_ts = new_ts.float() * (1000.0 / self.original_num_steps)
        return self.model(x, new_ts, **kwargs)
