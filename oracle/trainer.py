"""Oracle (test infrastructure): CPU restatement of the reference training step.

Follows /root/reference/tools/trainer.py: ema :12-18, sample_from_latent :21-25,
Trainer :28-150 (fp32 path: args.amp=False, no alignment encoder), and
/root/reference/tools/utils.py: warmup_cosine_lr/get_lr_lambda :75-90.
Pinned by tests/golden/trainer_*.json (5-step trajectories of the reference).
"""
import math
from contextlib import nullcontext

import torch
import torch.distributed as dist
import torch.nn as nn


def is_main_process():
    return (not dist.is_available()) or (not dist.is_initialized()) or dist.get_rank() == 0


def ema(source, target, decay):
    """Every state_dict entry (params AND buffers, frozen pos_embed included)."""
    with torch.no_grad():
        src, dst = source.state_dict(), target.state_dict()
        for k in src:
            dst[k].data.copy_(dst[k].data * decay + src[k].data * (1 - decay))


def sample_from_latent(latent, latent_scale=1.0):
    mean, std = torch.chunk(latent, 2, dim=1)
    return (mean + std * torch.randn_like(mean)) * latent_scale


def warmup_cosine_lr(step, warmup_steps, total_steps, lr, final_lr, cosine_decay):
    if step < warmup_steps:
        return min(step, warmup_steps) / warmup_steps
    if cosine_decay:
        prog = (step - warmup_steps) / (total_steps - warmup_steps)
        return (final_lr + (lr - final_lr) * 0.5 * (1 + math.cos(math.pi * prog))) / lr
    return 1


def get_lr_lambda(args):
    return lambda step: warmup_cosine_lr(step, args.warmup_steps, args.total_steps, args.lr, args.final_lr,
                                         args.cosine_decay)


class Trainer:
    def __init__(self, args, device, model, ema_model, optimizer, scheduler, diffusion, train_loader, pbar=None):
        assert not args.learn_align and not args.amp, "oracle covers the fp32, no-alignment path"
        self.args, self.device = args, device
        self.model, self.ema_model = model, ema_model
        self.optimizer, self.scheduler, self.diffusion = optimizer, scheduler, diffusion
        self.train_loader = train_loader
        self.datalooper = iter(train_loader)
        self.pbar = pbar

    def _get_next_batch(self):
        try:
            images, labels = next(self.datalooper)
        except StopIteration:
            self.datalooper = iter(self.train_loader)
            return self._get_next_batch()
        return images.to(self.device), (labels.to(self.device) if self.args.class_cond else None)

    def train_step(self, step):
        a = self.args
        self.model.train()
        if a.parallel:
            self.train_loader.sampler.set_epoch(step)
        accum = max(1, a.grad_accumulation)
        total, mse_avg = 0.0, 0.0
        for i in range(accum):
            images, labels = self._get_next_batch()
            if a.in_chans == 4:
                images = sample_from_latent(images, a.latent_scale)
            ctx = self.model.no_sync() if (a.parallel and accum > 1 and i < accum - 1) else nullcontext()
            with ctx:
                kw = {"y": labels} if a.class_cond else {}
                terms = self.diffusion.training_losses(self.model, images, None, model_kwargs=kw)
                loss = terms["loss"].mean() / accum
                loss.backward()
            total += loss.item()
            if "mse" in terms:                      # reference tools/trainer.py:116 (the KL objectives return only "loss")
                mse_avg += terms["mse"].mean().item() / accum
            if (i + 1) % accum == 0:
                if a.grad_clip:
                    nn.utils.clip_grad_norm_(self.model.parameters(), a.grad_clip)
                self.optimizer.step()
                self.optimizer.zero_grad()
        self.scheduler.step()
        if is_main_process():
            ema(self.model, self.ema_model, a.ema_decay)
            if self.pbar is not None:
                self.pbar.update(1)
                self.pbar.set_postfix(mse=f"{mse_avg:.4f}")
        self.last_mse = mse_avg
        return total
