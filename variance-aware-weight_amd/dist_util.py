"""Process-group helpers, same names as the reference's tools/dist_util.py:20-62.  One process per GPU;
backend "nccl" is RCCL on ROCm (xGMI between the 8 GPUs of a node); "gloo" when no GPU is present (CPU tests)."""
import os

import torch
import torch.distributed as dist


def is_main_process():
    if not dist.is_available() or not dist.is_initialized():
        return True
    return dist.get_rank() == 0


def dist_barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def setup_dist(backend=None, device_index=None):
    """device_index: GPU for this rank (default LOCAL_RANK).  A rehearsal of the N-rank path on a one-GPU box passes
    backend="gloo", device_index=0."""
    if dist.is_initialized():
        return
    local_rank = int(os.getenv("LOCAL_RANK", 0))
    use_gpu = torch.cuda.is_available()
    if use_gpu:
        torch.cuda.set_device(local_rank if device_index is None else device_index)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "12345")
    os.environ.setdefault("RANK", str(local_rank))
    os.environ.setdefault("WORLD_SIZE", str(torch.cuda.device_count() if use_gpu else 1))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group(backend=backend or ("nccl" if use_gpu else "gloo"), init_method="env://")


def cleanup_dist():
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


def dev():
    return torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")
