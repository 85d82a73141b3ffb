#!/usr/bin/env python3
"""Micro-benchmark of vaw_conv3x3 (implicit GEMM) on the conv shapes of a UNet workload (run on the GPU box).
    python tools/conv_bench.py [--shapes unet64|adm64] [--iters 10] [--modes 0,1,2]
Prints per-shape TFLOP/s (2*M*9*Ci*Co flop) of the forward, input-gradient and weight-gradient launches, HIP-event
timed, random operands."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vaw_amd  # noqa: E402,F401
from vaw_amd import ops  # noqa: E402
from vaw_amd._lib import BF16, ptr  # noqa: E402

SHAPES = {  # (B, H, Ci, Co)
    "small": [(128, 8, 384, 384), (128, 8, 768, 384), (128, 16, 384, 384), (64, 8, 384, 384), (32, 16, 384, 384)],
    "unet64": [(128, 64, 192, 192), (128, 32, 192, 384), (128, 32, 384, 384), (128, 16, 384, 384), (128, 8, 384, 384),
               (128, 16, 768, 384), (128, 32, 768, 384), (128, 64, 576, 192), (128, 64, 384, 192)],
    "adm64": [(256, 64, 192, 192), (256, 32, 192, 384), (256, 32, 384, 384), (256, 16, 384, 576), (256, 16, 576, 576),
              (256, 8, 576, 768), (256, 8, 768, 768), (256, 8, 1536, 768), (256, 16, 1152, 576)],
}


def run(B, H, Ci, Co, mode, iters):
    dev = "cuda"
    M = B * H * H
    x = torch.randn(M, Ci, device=dev).bfloat16()
    dy = torch.randn(M, Co, device=dev).bfloat16()
    w = torch.randn(Co, 9 * Ci, device=dev).bfloat16()
    bias = torch.randn(Co, device=dev)
    if mode == 0:
        out = torch.empty(M, Co, device=dev, dtype=torch.bfloat16)
        call = lambda: ops.conv3x3(BF16, 0, ptr(x), None, ptr(w), ptr(out), B, H, H, Ci, Co, bias=ptr(bias))
    elif mode == 1:
        out = torch.empty(M, Ci, device=dev, dtype=torch.bfloat16)
        call = lambda: ops.conv3x3(BF16, 1, ptr(dy), None, ptr(w), ptr(out), B, H, H, Ci, Co)
    else:
        out = torch.zeros(Co, 9 * Ci, device=dev)
        call = lambda: ops.conv3x3(BF16, 2, ptr(dy), ptr(x), None, ptr(out), B, H, H, Ci, Co, beta=1.0)
    for _ in range(2):
        assert call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        call()
    e1.record()
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / iters
    name = ("fwd", "dgrad", "wgrad")[mode]
    print(f"B={B} {H:3d}x{H:<3d} Ci={Ci:5d} Co={Co:5d} {name:6s} {us:9.1f} us  {2.0 * M * 9 * Ci * Co / us / 1e6:8.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="unet64")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--modes", default="0,1,2")
    ap.add_argument("--tiles", default="-1", help="comma list of vaw_debug_gemm_tile modes: -1 by shape, 0 128x128 kernel, 2 / 3 persistent 256 / 192 columns")
    a = ap.parse_args()
    from vaw_amd._lib import lib
    for (B, H, Ci, Co) in SHAPES[a.shapes]:
        for m in [int(v) for v in a.modes.split(",")]:
            for t in [int(v) for v in a.tiles.split(",")]:
                lib().vaw_debug_gemm_tile(t)
                print(f"[tile {t:2d}] ", end="")
                run(B, H, Ci, Co, m, a.iters)
