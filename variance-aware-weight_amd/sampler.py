"""Classifier-free guidance wrapper of the sampling side: same class as the reference's tools/sampler.py:10-48.
(The rest of that file -- VAE decode, classifier guidance, EDM / flow samplers, sample gathering -- is out of scope.)"""
import torch


class IntervalCFG(torch.nn.Module):
    """out = uncond + s * (cond - uncond) when the (mean) timestep lies in `interval` (or always, when the interval is
    not a valid range); the unconditional half uses the null label `num_classes`; one doubled-batch model call."""

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
        time_from, time_to = self.interval
        return time_from <= time_value < time_to if time_from >= 0 and time_to > time_from else True

    def _format_time(self, time_tensor, batch_size):
        if time_tensor.dim() == 0:
            return time_tensor.expand(batch_size)
        if time_tensor.numel() == 1:
            return time_tensor.reshape(1).expand(batch_size)
        return time_tensor.reshape(batch_size)

    def forward(self, sample_tensor, time_tensor, **model_kwargs):
        time_tensor = self._format_time(time_tensor, sample_tensor.shape[0])
        class_labels = model_kwargs.get("y", None)
        if not (self.class_cond and class_labels is not None and self._use_cfg(float(time_tensor.float().mean().item()))):
            return self.model(sample_tensor, time_tensor, **model_kwargs)
        assert class_labels.shape[0] == sample_tensor.shape[0], \
            f"CFG expects label batch size {sample_tensor.shape[0]}, but got {class_labels.shape[0]}."
        cfg_kwargs = dict(model_kwargs)
        cfg_kwargs["y"] = torch.cat([class_labels, torch.full_like(class_labels, self.null_label)], dim=0)
        model_output = self.model(torch.cat([sample_tensor, sample_tensor], dim=0), torch.cat([time_tensor, time_tensor], dim=0),
                                  **cfg_kwargs)
        model_output = model_output[0] if isinstance(model_output, tuple) else model_output
        cond_output, uncond_output = model_output.chunk(2, dim=0)
        return uncond_output + self.guidance_scale * (cond_output - uncond_output)
