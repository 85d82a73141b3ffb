"""Interval classifier-free guidance for the sampling side (behaviour of the reference's tools/sampler.py:10-48, pinned
by tests/golden/sampling.pt).  The remaining samplers of that file live in samplers.py; VAE decode and classifier
guidance are out of scope."""
import torch


class IntervalCFG(torch.nn.Module):
    """Guided denoiser:  eps = eps(null) + s * (eps(y) - eps(null)).

    Guidance is active when s != 1, labels are given, and -- if `interval` = (lo, hi) is a proper range with lo >= 0 --
    the batch's mean timestep lies in [lo, hi).  Active calls evaluate the wrapped model ONCE on the batch stacked on
    itself, second half labelled with the null class `num_classes`; inactive calls pass straight through."""

    def __init__(self, model, num_classes, guidance_scale=1.0, interval=(-1.0, -1.0), class_cond=True):
        super().__init__()
        self.model = model
        self.null_label = int(num_classes)
        self.guidance_scale = float(guidance_scale)
        self.interval = interval
        self.class_cond = class_cond

    def guidance_active(self, t_mean):
        """Pure host-side predicate on the mean timestep of a call."""
        if abs(self.guidance_scale - 1.0) < 1e-8:
            return False
        lo, hi = self.interval
        bounded = lo >= 0 and hi > lo
        return (lo <= t_mean < hi) if bounded else True

    def forward(self, x, t, **model_kwargs):
        n = x.shape[0]
        t = t.reshape(-1)
        if t.numel() == 1:
            t = t.expand(n)
        elif t.numel() != n:
            raise ValueError(f"IntervalCFG: {t.numel()} timesteps for a batch of {n}")
        y = model_kwargs.get("y")
        guided = self.class_cond and y is not None and self.guidance_active(float(t.float().mean()))
        if not guided:
            return self.model(x, t, **model_kwargs)
        if y.shape[0] != n:
            raise AssertionError(f"CFG expects label batch size {n}, but got {y.shape[0]}.")
        stacked = {**model_kwargs, "y": torch.cat((y, y.new_full(y.shape, self.null_label)))}
        out = self.model(x.repeat(2, *([1] * (x.dim() - 1))), t.repeat(2), **stacked)
        if isinstance(out, tuple):          # DiT returns (eps, aux)
            out = out[0]
        with_label, without = out[:n], out[n:]
        return without + self.guidance_scale * (with_label - without)
