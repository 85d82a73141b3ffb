"""Run utilities on the training path, same names as the reference's tools/utils.py:
set_random_seed :62-72, warmup_cosine_lr / get_lr_lambda :75-90, save_checkpoint / load_checkpoint :93-120."""
import argparse
import math
import os
import random

import numpy as np
import torch
import torch.distributed as dist

from . import dist_util


def str2bool(v):
    if isinstance(v, bool):
        return v
    if v.lower() in ("yes", "true", "t", "y", "1"):
        return True
    if v.lower() in ("no", "false", "f", "n", "0"):
        return False
    raise argparse.ArgumentTypeError("Boolean value expected.")


def set_random_seed(args, seed):
    """seed + rank, so every rank draws different t / noise (reference :62-69)."""
    rank = dist.get_rank() if (args.parallel and dist.is_initialized()) else 0
    seed = seed + rank
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def warmup_cosine_lr(step, warmup_steps, total_steps, lr, final_lr, cosine_decay):
    if step < warmup_steps:
        return min(step, warmup_steps) / warmup_steps
    if cosine_decay:
        progress = (step - warmup_steps) / (total_steps - warmup_steps)
        return (final_lr + (lr - final_lr) * 0.5 * (1 + math.cos(math.pi * progress))) / lr
    return 1


def get_lr_lambda(args):
    return lambda step: warmup_cosine_lr(step, args.warmup_steps, args.total_steps, args.lr, args.final_lr,
                                         args.cosine_decay)


def _strip_module(sd):
    return {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}


def save_checkpoint(args, step, model, optimizer, ema_model=None, scheduler=None):
    """{'model','optimizer','step','ema_model'} with the reference's state_dict keys (a wrapped model saves
    'module.'-prefixed keys, as torch DDP does).  'scheduler' is an addition the reference forgets."""
    zero = getattr(optimizer, "zero", None)
    opt_state = None
    if zero is not None:       # sharded optimizer (ZeRO-1): collectives -- EVERY rank calls save_checkpoint; rank 0 writes the file
        optimizer.consolidate()
        if getattr(optimizer, "_ema_shard", None) is not None:
            optimizer.consolidate_ema(ema_model if dist_util.is_main_process() else None)
        opt_state = optimizer.state_dict()
    if not dist_util.is_main_process():
        return None
    d = os.path.join(args.logdir, "checkpoint")
    os.makedirs(d, exist_ok=True)
    state = {"model": model.state_dict(), "optimizer": opt_state if opt_state is not None else optimizer.state_dict(), "step": step}
    if ema_model is not None:
        state["ema_model"] = ema_model.state_dict()
    if scheduler is not None:
        state["scheduler"] = scheduler.state_dict()
    path = os.path.join(d, f"{args.model}_{args.mean_type}_{args.path_type}_{step}.pth")
    torch.save(state, path)
    return path


def load_checkpoint(ckpt_path, model=None, optimizer=None, ema_model=None, scheduler=None):
    assert os.path.exists(ckpt_path), f"Error: checkpoint {ckpt_path} not found"
    ck = torch.load(ckpt_path, map_location="cpu", weights_only=True)

    def load_into(m, sd):
        wrapped = hasattr(m, "module")
        has_prefix = any(k.startswith("module.") for k in sd)
        if wrapped and not has_prefix:
            sd = {"module." + k: v for k, v in sd.items()}
        elif not wrapped and has_prefix:
            sd = _strip_module(sd)
        m.load_state_dict(sd)

    if model is not None:
        load_into(model, ck["model"])
    if optimizer is not None:
        optimizer.load_state_dict(ck["optimizer"])
    if ema_model is not None and "ema_model" in ck:
        load_into(ema_model, ck["ema_model"])
    if scheduler is not None and "scheduler" in ck:
        scheduler.load_state_dict(ck["scheduler"])
    return ck
