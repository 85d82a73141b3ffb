"""Oracle (test infrastructure): classifier-free guidance wrapper, following /root/reference/tools/sampler.py:10-48
(IntervalCFG).  Pinned by tests/golden/sampling.pt."""
import torch


class IntervalCFG(torch.nn.Module):
    def __init__(self, model, num_classes, guidance_scale=1.0, interval=(-1.0, -1.0), class_cond=True):
        super().__init__()
        self.model = model
        self.null_label = int(num_classes)
        self.guidance_scale = float(guidance_scale)
        self.interval = interval
        self.class_cond = class_cond

    def _use_cfg(self, time_value):
        if abs(self.guidance_scale - 1.0) < 1e-8:
            return False
        lo, hi = self.interval
        return lo <= time_value < hi if lo >= 0 and hi > lo else True

    def _format_time(self, t, batch):
        if t.dim() == 0:
            return t.expand(batch)
        if t.numel() == 1:
            return t.reshape(1).expand(batch)
        return t.reshape(batch)

    def forward(self, x, t, **model_kwargs):
        t = self._format_time(t, x.shape[0])
        y = model_kwargs.get("y", None)
        if not (self.class_cond and y is not None and self._use_cfg(float(t.float().mean().item()))):
            return self.model(x, t, **model_kwargs)
        assert y.shape[0] == x.shape[0]
        kw = dict(model_kwargs)
        kw["y"] = torch.cat([y, torch.full_like(y, self.null_label)], dim=0)
        out = self.model(torch.cat([x, x], dim=0), torch.cat([t, t], dim=0), **kw)
        out = out[0] if isinstance(out, tuple) else out
        cond, uncond = out.chunk(2, dim=0)
        return uncond + self.guidance_scale * (cond - uncond)
