#!/usr/bin/env python3
"""Micro-benchmark of the HBM-bound kernels of the DiT step at its real sizes, TB/s of ALGORITHMIC bytes (run on the GPU box).
    python tools/hbm_bench.py [--iters 20] [--B 256 --T 64 --D 768 --H 12] [--params 130400000] [--only adamw,rowbwd,...]
Every kernel runs over rotating buffer sets so that nothing is served from the 256 MiB Infinity Cache between iterations
(in the step each of these tensors was last touched by another kernel hundreds of MB ago)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vaw_amd  # noqa: E402,F401
from vaw_amd import ops  # noqa: E402
from vaw_amd._lib import BF16, ptr  # noqa: E402


def timeit(fns, iters):
    """fns: list of closures over distinct buffer sets, called round-robin"""
    for f in fns:
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for i in range(iters):
        fns[i % len(fns)]()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / iters


def report(name, us, nbytes):
    print(f"{name:28s} {us:9.1f} us  {nbytes / 1e6:9.1f} MB  {nbytes / us / 1e6:6.2f} TB/s", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--B", type=int, default=256)
    ap.add_argument("--T", type=int, default=64)
    ap.add_argument("--D", type=int, default=768)
    ap.add_argument("--H", type=int, default=12)
    ap.add_argument("--params", type=int, default=130_400_000)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    only = set(a.only.split(",")) if a.only else None
    want = lambda k: only is None or k in only
    dev = "cuda"
    B, T, D, H = a.B, a.T, a.D, a.H
    M = B * T
    NSET = max(2, int(400e6 // (M * D * 4)) + 1)         # rotate over > 256 MiB of the largest operand

    if want("adamw"):
        n = a.params
        p, g, m, v, e = (torch.randn(n, device=dev) * 0.01 for _ in range(5))
        v.abs_()
        sh = torch.empty(n, device=dev, dtype=torch.bfloat16)
        ss = torch.zeros(1, device=dev)
        us = timeit([lambda: ops.adamw_ema_step(p, g, m, v, e, sh, 1e-4, 0.9, 0.999, 1e-8, 0.0, 3, 0.9999, ss, None, False)], a.iters)
        report("adamw+ema+shadow (38 B/param)", us, 38.0 * n)
        del p, g, m, v, e, sh

    mod = torch.randn(B, 6 * D, device=dev)
    if want("lnfwd"):
        sets = []
        for _ in range(NSET):
            x = torch.randn(M, D, device=dev)
            out = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
            mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
            sets.append((x, out, mean, rstd))
        fns = [(lambda s=s: ops.ln_modulate_fwd(BF16, ptr(s[0]), ptr(mod), ptr(mod) + 4 * D, 6 * D, ptr(s[1]), ptr(s[2]), ptr(s[3]), B, T, D)) for s in sets]
        report("ln_modulate_fwd (6 B/elem)", timeit(fns, a.iters), 6.0 * M * D)
        del sets, fns

    if want("rowbwd"):
        sets = []
        dmod = torch.zeros(B, 6 * D, device=dev)
        colp = torch.zeros(B, D, device=dev)
        for _ in range(NSET):
            s = dict(dout=torch.randn(M, D, device=dev).bfloat16(), x=torch.randn(M, D, device=dev), mean=torch.randn(M, device=dev),
                     rstd=torch.rand(M, device=dev) + 0.5, dres=torch.randn(M, D, device=dev), dx=torch.empty(M, D, device=dev),
                     y=torch.randn(M, D, device=dev).bfloat16(), dy=torch.empty(M, D, device=dev, dtype=torch.bfloat16))
            sets.append(s)
        fns = [(lambda s=s: ops.ln_modulate_bwd_gate(BF16, ptr(s["dout"]), ptr(s["x"]), ptr(s["mean"]), ptr(s["rstd"]), ptr(mod) + 4 * D, 6 * D,
                                                     ptr(s["dres"]), ptr(s["dx"]), ptr(dmod), ptr(dmod) + 4 * D, 6 * D, ptr(s["y"]),
                                                     ptr(mod) + 8 * D, ptr(s["dy"]), ptr(dmod) + 8 * D, B, T, D, ptr(colp))) for s in sets]
        report("ln_bwd + gate_bwd fused (18 B/elem)", timeit(fns, a.iters), 18.0 * M * D)
        fns = [(lambda s=s: ops.ln_modulate_bwd(BF16, ptr(s["dout"]), ptr(s["x"]), ptr(s["mean"]), ptr(s["rstd"]), ptr(mod) + 4 * D, 6 * D,
                                                ptr(s["dres"]), ptr(s["dx"]), ptr(dmod), ptr(dmod) + 4 * D, 6 * D, B, T, D)) for s in sets]
        report("ln_bwd alone (14 B/elem)", timeit(fns, a.iters), 14.0 * M * D)
        del sets, fns

    if want("attn"):
        hd = D // H
        sets = []
        for _ in range(NSET):
            qkv = (torch.randn(M, 3 * D, device=dev) * 0.5).bfloat16()
            do = torch.randn(M, D, device=dev).bfloat16()
            o = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
            lse, delta = torch.empty(B * H * T, device=dev), torch.empty(B * H * T, device=dev)
            dqkv = torch.empty_like(qkv)
            sets.append((qkv, do, o, lse, delta, dqkv))
        desc = ops.attn_desc_token_major(B, H, T, hd)
        es = 2
        fw = [(lambda s=s: ops.attn_fwd(BF16, desc, ptr(s[0]), ptr(s[0]) + es * D, ptr(s[0]) + 2 * es * D, ptr(s[2]), ptr(s[3]))) for s in sets]
        report("attn fwd (8 B/elem of [M,D])", timeit(fw, a.iters), 8.0 * M * D)
        part = ops.ColsumPartial(M // 64 + 8, 3 * D, torch.device(dev)) if hasattr(ops, "ColsumPartial") else None
        bw = [(lambda s=s: ops.attn_bwd(BF16, desc, ptr(s[0]), ptr(s[0]) + es * D, ptr(s[0]) + 2 * es * D, ptr(s[2]), ptr(s[1]), ptr(s[3]),
                                        ptr(s[4]), ptr(s[5]), ptr(s[5]) + es * D, ptr(s[5]) + 2 * es * D)) for s in sets]
        report("attn bwd (16 B/elem of [M,D])", timeit(bw, a.iters), 16.0 * M * D)
        if part is not None:
            bwc = [(lambda s=s: ops.attn_bwd_colsum(BF16, desc, ptr(s[0]), ptr(s[0]) + es * D, ptr(s[0]) + 2 * es * D, ptr(s[2]), ptr(s[1]),
                                                    ptr(s[3]), ptr(s[4]), ptr(s[5]), ptr(s[5]) + es * D, ptr(s[5]) + 2 * es * D, part)) for s in sets]
            report("attn bwd + column sums", timeit(bwc, a.iters), 16.0 * M * D)


if __name__ == "__main__":
    main()
