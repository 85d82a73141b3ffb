#!/usr/bin/env python3
"""Micro-benchmark of vaw_gemm on the shapes of a workload (run on the GPU box).
    python tools/gemm_bench.py [--shapes dit_b4|square] [--iters 20]
Prints per-shape TFLOP/s for the forward (k-major x k-major), dgrad (k-major x mn-major) and wgrad
(mn-major x mn-major) layouts, HIP-event timed, operands random (zeros would flatter the clock)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vaw_amd  # noqa: E402
from vaw_amd import ops  # noqa: E402
from vaw_amd._lib import BF16, ptr  # noqa: E402

M = 16384
DIT_B4 = [  # (name, layout, M, N, K, epilogue)
    ("qkv fwd", "fwd", M, 2304, 768, "bias"), ("proj fwd", "fwd", M, 768, 768, "gate"), ("fc1 fwd", "fwd", M, 3072, 768, "gelu"),
    ("fc2 fwd", "fwd", M, 768, 3072, "gate"),
    ("qkv dgrad", "dgrad", M, 768, 2304, "none"), ("proj dgrad", "dgrad", M, 768, 768, "none"),
    ("fc1 dgrad", "dgrad", M, 768, 3072, "none"), ("fc2 dgrad", "dgrad", M, 3072, 768, "dgelu"),
    ("qkv wgrad", "wgrad", 2304, 768, M, "f32"), ("proj wgrad", "wgrad", 768, 768, M, "f32"),
    ("fc1 wgrad", "wgrad", 3072, 768, M, "f32"), ("fc2 wgrad", "wgrad", 768, 3072, M, "f32"),
]
UNET64 = [  # a few conv-as-GEMM shapes of UNet_64 at batch 128 (M = B*H*W pixels, K = 9*Ci)
    ("c192@64 fwd", "fwd", 128 * 64 * 64, 192, 9 * 192, "bias"), ("c384@32 fwd", "fwd", 128 * 32 * 32, 384, 9 * 384, "bias"),
    ("c576@16 fwd", "fwd", 128 * 16 * 16, 576, 9 * 576, "bias"), ("c768@8 fwd", "fwd", 128 * 8 * 8, 768, 9 * 768, "bias"),
]
# per-round cost of the persistent kernel: 256 x 256 tiles, 8 column tiles, 32 / 64 / 96 / 128 row tiles = 1..4 rounds of 256
ROUNDS = [(f"r{r} K768", "fwd", 8192 * r, 2048, 768, "none") for r in (1, 2, 3, 4)] + \
         [(f"r{r} K768 gelu", "fwd", 8192 * r, 2048, 768, "gelu") for r in (1, 2, 3)] + \
         [(f"r{r} K3072", "fwd", 8192 * r, 2048, 3072, "none") for r in (1, 2)]
SQUARE = [("4096^3 fwd", "fwd", 4096, 4096, 4096, "none"), ("4096^3 dgrad", "dgrad", 4096, 4096, 4096, "none"),
          ("4096^3 wgrad", "wgrad", 4096, 4096, 4096, "f32"), ("8192^3 fwd", "fwd", 8192, 8192, 8192, "none")]


def run(name, layout, M, N, K, epi, iters):
    dev = "cuda"
    ak, bk = {"fwd": (1, 1), "dgrad": (1, 0), "wgrad": (0, 0)}[layout]
    A = torch.randn((M, K) if ak else (K, M), device=dev).bfloat16()
    B = torch.randn((N, K) if bk else (K, N), device=dev).bfloat16()
    out_f32 = epi in ("f32", "gate")
    C = torch.empty(M, N, device=dev, dtype=torch.float32 if out_f32 else torch.bfloat16)
    kw = {}
    keep = []
    if epi in ("bias", "gelu", "gate"):
        b = torch.randn(N, device=dev); keep.append(b); kw["bias"] = ptr(b)
    if epi == "gelu":
        aux = torch.empty(M, N, device=dev, dtype=torch.bfloat16); keep.append(aux); kw.update(act=1, aux_out=ptr(aux))
    if epi == "dgelu":
        aux = torch.randn(M, N, device=dev).bfloat16(); keep.append(aux); kw.update(act=2, aux_in=ptr(aux))
    if epi == "gate":
        g = torch.randn(M // 64, N, device=dev); r = torch.randn(M, N, device=dev)
        aux = torch.empty(M, N, device=dev, dtype=torch.bfloat16); keep += [g, r, aux]
        kw.update(gate=ptr(g), gate_ld=N, resid=ptr(r), rows_per_batch=64, aux_out=ptr(aux))
    lda, ldb = A.shape[1], B.shape[1]
    call = lambda: ops.gemm(BF16, ak, bk, M, N, K, ptr(A), lda, ptr(B), ldb, ptr(C), N, out_f32=out_f32, **kw)
    for _ in range(3):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        call()
    e1.record()
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / iters
    print(f"{name:14s} {layout:6s} M={M:6d} N={N:5d} K={K:6d} epi={epi:6s} {us:9.1f} us  {2.0 * M * N * K / us / 1e6:8.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="dit_b4")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", default=None, help="substring of the shape name")
    ap.add_argument("--tile", default="auto", help="auto | 128 | 256 | both | p8 | pd (auto vs parked-drain) | pdcmp | cmp (128 vs persistent) | sm (dispatcher without the "
                                                   "small-M kernel vs gemm_sm with 64- and 128-column tiles): bf16 MFMA kernel")
    ap.add_argument("--m", type=int, default=None, help="token rows instead of 16384 (dit_b4 forward / input-gradient shapes)")
    a = ap.parse_args()
    if a.tile == "sm":
        os.environ["VAW_SM_MAX_M"] = "0"        # "auto" then is the dispatcher as it was before gemm_sm.hip (read once, at the first launch)
    if a.m:
        DIT_B4[:] = [(n, l, a.m, N, K, e) for (n, l, _, N, K, e) in DIT_B4 if l != "wgrad"]
    from vaw_amd._lib import lib
    tiles = {"auto": [-1], "128": [0], "256": [1], "both": [0, 1], "p8": [4], "cmp": [0, 2, 3], "p8_256": [2], "p8_192": [3], "sm": [-1, 6, 7, 8], "pd": [-1, 9], "pdcmp": [2, 3, 10, 11], "ws": [2, 3, 13, 14]}[a.tile]
    for row in {"dit_b4": DIT_B4, "square": SQUARE, "unet64": UNET64, "all": DIT_B4 + SQUARE, "rounds": ROUNDS}[a.shapes]:
        if a.only is None or a.only in row[0]:
            for t in tiles:
                lib().vaw_debug_gemm_tile(t)
                if len(tiles) > 1:
                    print(f"[tile {('128', '256', 'p8/256', 'p8/192', 'p8', 'sm', 'sm/64', 'sm/128', 'sm/128x128', 'pd', 'pd/256', 'pd/192', 'ws', 'ws/256', 'ws/192', 'auto')[t]}] ", end="")
                run(*row, a.iters)
