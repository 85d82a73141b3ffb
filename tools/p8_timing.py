#!/usr/bin/env python3
"""Where the cycles of a K-tile phase of the persistent GEMM go (run on the GPU box after `make -C variance-aware-weight_amd/csrc timing`):
    VAW_HIP_LIB=variance-aware-weight_amd/libvaw_hip_timing.so python tools/p8_timing.py [M N K]
The timing build accumulates s_memtime deltas in wave 0 (upper wave row) and wave 4 (lower wave row) of workgroup 0 between the five
points of each phase: start -> fragments + DMA issued -> vmcnt wait done -> barrier 1 passed -> MFMAs issued -> (barrier 2) -> next start.
s_memtime ticks at 100 MHz on this part; the table reports ns per phase."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vaw_amd  # noqa: E402,F401
from vaw_amd import ops  # noqa: E402
from vaw_amd._lib import BF16, lib, ptr  # noqa: E402

M, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (4096, 4096, 4096)
A = torch.randn(M, K, device="cuda").bfloat16()
B = torch.randn(N, K, device="cuda").bfloat16()
C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
lib().vaw_debug_gemm_tile(2)      # persistent kernel, 256-column tiles
for _ in range(3):
    ops.gemm(BF16, 1, 1, M, N, K, ptr(A), K, ptr(B), K, ptr(C), N)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
ops.gemm(BF16, 1, 1, M, N, K, ptr(A), K, ptr(B), K, ptr(C), N)
e1.record()
torch.cuda.synchronize()
us = 1e3 * e0.elapsed_time(e1)
print(f"{M}x{N}x{K}: {us:.1f} us, {2.0 * M * N * K / us / 1e6:.1f} TFLOP/s")
raw = C.view(torch.int16).cpu()
names = ["issue (frags+DMA)", "vmcnt wait", "barrier 1", "MFMA issue", "barrier 2", "phases"]
for w, row in ((0, 0), (4, 128)):
    t = raw[row, :24].contiguous().view(torch.int64).tolist()
    n = max(t[5], 1)
    print(f"wave {w}: " + ", ".join(f"{nm} {10.0 * v / n:.1f} ns" for nm, v in zip(names[:5], t[:5])) + f"  ({t[5]} phases, sum {10.0 * sum(t[:5]) / n:.1f} ns/phase)")
