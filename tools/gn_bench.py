#!/usr/bin/env python3
"""Micro-benchmark of vaw_groupnorm_fwd / vaw_groupnorm_bwd (NHWC, 32 groups, SiLU + FiLM) against the HBM roofline, streaming
kernels and cooperative single-read kernels (vaw_debug_gn_coop 0 / 1).  The TB/s column prices both at the STREAMING kernels' traffic
(forward = x twice + y: 3 passes; backward = dy, x twice + dx: 5 passes) so the two lines of a shape compare directly; the
cooperative kernels move 2 and 3 passes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vaw_amd  # noqa: E402,F401
from vaw_amd import ops  # noqa: E402
from vaw_amd._lib import BF16, lib, ptr, stream_ptr  # noqa: E402

for (B, HW, C) in [(256, 4096, 192), (256, 1024, 384), (256, 256, 576), (256, 64, 768), (128, 4096, 192)]:
    M = B * HW
    x = torch.randn(M, C, device="cuda").bfloat16()
    dy = torch.randn(M, C, device="cuda").bfloat16()
    y, dx = torch.empty_like(x), torch.empty_like(x)
    gam, bet = torch.randn(C, device="cuda"), torch.randn(C, device="cuda")
    film = torch.randn(B, 2 * C, device="cuda")
    mean, rstd = torch.empty(B * 32, device="cuda"), torch.empty(B * 32, device="cuda")
    dg, db, dfilm = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda"), torch.zeros(B, 2 * C, device="cuda")
    ws = ops.scratch_f32(torch.device("cuda", 0), lib().vaw_groupnorm_workspace_floats(B, HW, C))
    fwd = lambda: lib().vaw_groupnorm_fwd(BF16, ptr(x), ptr(gam), ptr(bet), ptr(film), ptr(film) + 4 * C, 2 * C, 1, ptr(y), ptr(mean),
                                          ptr(rstd), B, HW, C, 32, 1e-5, ptr(ws), stream_ptr())
    bwd = lambda: lib().vaw_groupnorm_bwd(BF16, ptr(dy), ptr(x), ptr(mean), ptr(rstd), ptr(gam), ptr(bet), ptr(film), ptr(film) + 4 * C,
                                          2 * C, 1, None, ptr(dx), ptr(dg), ptr(db), 0.0, ptr(dfilm), ptr(dfilm) + 4 * C, 2 * C, B, HW, C, 32,
                                          ptr(ws), stream_ptr())
    for coop, fn, nm, passes in ((0, fwd, "fwd", 3), (1, fwd, "fwd", 3), (0, bwd, "bwd", 5), (1, bwd, "bwd", 5)):
        lib().vaw_debug_gn_coop(coop)
        nm = nm + ("/coop" if coop else "/stream")
        for _ in range(2):
            assert fn() == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = 1e2 * e0.elapsed_time(e1)
        print(f"B={B} HW={HW:5d} C={C:4d} {nm:10s} {us:8.1f} us  {passes * M * C * 2 / us / 1e6:6.2f} TB/s if {passes} passes", flush=True)
