#!/usr/bin/env python3
"""One launch of the parked-drain GEMM on one shape against torch (bisecting aid; run on the GPU box).
    python tools/pd_probe.py M N K [fwd|dgrad] [tile] [bias|nobias] [check|nocheck]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vaw_amd  # noqa
from vaw_amd import ops
from vaw_amd._lib import lib

M, N, K = (int(v) for v in sys.argv[1:4])
layout = sys.argv[4] if len(sys.argv) > 4 else "fwd"
tile = int(sys.argv[5]) if len(sys.argv) > 5 else 10
bias = (sys.argv[6] if len(sys.argv) > 6 else "bias") == "bias"
check = (sys.argv[7] if len(sys.argv) > 7 else "check") == "check"
colsum = (sys.argv[8] if len(sys.argv) > 8 else "") == "colsum"
g = torch.Generator().manual_seed(1)
A = torch.randint(-3, 4, (M, K), generator=g).float()
B = torch.randint(-3, 4, (N, K) if layout == "fwd" else (K, N), generator=g).float()
b = torch.randint(-3, 4, (N,), generator=g).float() if bias else None
ref = A.double() @ (B.double().t() if layout == "fwd" else B.double())
if bias:
    ref = ref + b.double()
ref = ref.float().bfloat16()
Ad, Bd = A.cuda().bfloat16(), B.cuda().bfloat16()
lib().vaw_debug_gemm_tile(tile)
cs = torch.zeros(N, device="cuda") if colsum else None
out = ops.gemm_t(Ad, Bd, a_kmajor=True, b_kmajor=layout == "fwd", bias=b.cuda() if bias else None, colsum_out=cs)
torch.cuda.synchronize()
lib().vaw_debug_gemm_tile(-1)
if check:
    if colsum:
        err = (cs.cpu().double() - ref.double().sum(0)).abs().max().item()
        print('colsum max err', err)
        assert err < 1e-2
    bad = (out.cpu() != ref)
    print(f"pd_probe {M}x{N}x{K} {layout} tile {tile} bias={bias}: mismatches {int(bad.sum())} of {bad.numel()}")
    if bad.any():
        idx = bad.nonzero()
        print("first bad:", idx[:8].tolist(), "rows with errors:", sorted(set((idx[:, 0] // 16).tolist()))[:20], "cols/16:", sorted(set((idx[:, 1] // 16).tolist()))[:20])
        sys.exit(1)
else:
    print(f"pd_probe {M}x{N}x{K} {layout} tile {tile}: ran")
