"""Timestep respacing for the sampling side: same names and behaviour as the reference's tools/respace.py
(space_timesteps :8-62, SpacedDiffusion :65-112; its model wrapper :115-130 is a closure here) over the HIP-backed GaussianDiffusion."""
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
        return super()._reverse_step(kind, self._respaced(model), *args, **kwargs)

    def training_losses(self, model, *args, **kwargs):
        return super().training_losses(self._respaced(model), *args, **kwargs)

    def _scale_timesteps(self, t):
        return t            # the respaced call (below) hands the model original, already rescaled timesteps

    def _kept_steps_on(self, like):
        """The kept original timesteps k_j as a tensor living beside `like` (one upload per device / index dtype)."""
        cache = self.__dict__.setdefault("_kept_cache", {})
        key = (like.device, like.dtype)
        if key not in cache:
            cache[key] = torch.as_tensor(self.timestep_map, device=like.device, dtype=like.dtype)
        return cache[key]

    def _respaced(self, model):
        """Callable with the model's signature that receives indices j of the SHORT chain and evaluates the model at the
        ORIGINAL timestep k_j (times 1000 / T_original when rescale_timesteps), what the reference's `_WrappedModel`
        (tools/respace.py:115-130) does.  A model wrapped once is passed through unchanged."""
        if getattr(model, "_vaw_respaced_by", None) is self:
            return model
        factor = 1000.0 / self.original_num_steps if self.rescale_timesteps else None

        def call(x, j, **kwargs):
            k = self._kept_steps_on(j).index_select(0, j.reshape(-1)).reshape(j.shape)
            return model(x, k.float() * factor if factor is not None else k, **kwargs)

        call._vaw_respaced_by = self
        if hasattr(model, "parameters"):
            call.parameters = model.parameters      # GaussianDiffusion asks the model for its device through this
        return call
