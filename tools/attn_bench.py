#!/usr/bin/env python3
"""Micro-benchmark of vaw_attn_fwd / vaw_attn_bwd (token-major qkv layout) on the attention shapes of the workloads.
    python tools/attn_bench.py [--iters 10]
TFLOP/s counts 4*T^2*hd*H*B forward and 2.5x that backward (the recomputation inside the backward is not counted)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vaw_amd  # noqa: E402,F401
from vaw_amd import ops  # noqa: E402
from vaw_amd._lib import ptr  # noqa: E402

SHAPES = [("DiT-B/4", 256, 12, 64, 64), ("DiT-B/2", 256, 12, 256, 64), ("DiT-XL/2", 128, 16, 256, 72),
          ("ADM_64 32x32", 256, 6, 1024, 64), ("ADM_64 16x16", 256, 9, 256, 64), ("UNet_64 16x16", 128, 4, 256, 96)]


def run(name, B, H, T, hd, iters):
    D = H * hd
    qkv = (torch.randn(B * T, 3 * D, device="cuda") * 0.5).bfloat16()
    do = torch.randn(B * T, D, device="cuda").bfloat16()
    o = torch.empty(B * T, D, device="cuda", dtype=torch.bfloat16)
    lse, delta = torch.empty(B * H * T, device="cuda"), torch.empty(B * H * T, device="cuda")
    dqkv = torch.empty_like(qkv)
    desc = ops.attn_desc_token_major(B, H, T, hd)
    dt, es = ops.dt_of(o), 2
    fwd = lambda: ops.attn_fwd(dt, desc, ptr(qkv), ptr(qkv) + es * D, ptr(qkv) + 2 * es * D, ptr(o), ptr(lse))
    bwd = lambda: ops.attn_bwd(dt, desc, ptr(qkv), ptr(qkv) + es * D, ptr(qkv) + 2 * es * D, ptr(o), ptr(do), ptr(lse), ptr(delta),
                               ptr(dqkv), ptr(dqkv) + es * D, ptr(dqkv) + 2 * es * D)
    flop = 4.0 * T * T * hd * H * B
    for fn, nm, f in ((fwd, "fwd", flop), (bwd, "bwd", 2.5 * flop)):
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / iters
        print(f"{name:14s} B={B:4d} H={H:3d} T={T:5d} hd={hd:4d} {nm} {us:9.1f} us  {f / us / 1e6:7.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    for row in SHAPES:
        run(*row, a.iters)
