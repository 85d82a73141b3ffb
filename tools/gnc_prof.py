#!/usr/bin/env python3
"""Stage timeline of one workgroup of the cooperative GroupNorm forward (library built with -DGNC_PROF=1: make exp XN=gncprof
XF=-DGNC_PROF=1 XSRC=unet_ops; run with VAW_HIP_LIB=.../libvaw_hip_gncprof.so).  Stamps are wall_clock64 (100 MHz) taken by lane 0
of wave 1 of workgroup 7: 1 wait begins, 2 partners arrived, 3 partials fetched, 4 statistics ready, 5 phase 1 begins, 6 sums
done (loads consumed), 7 chunk folded, 8 published, 9 phase 2 begins, 10 phase 2 stores issued."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vaw_amd  # noqa: E402,F401
from vaw_amd import ops  # noqa: E402
from vaw_amd._lib import BF16, lib, ptr, stream_ptr  # noqa: E402

B, HW, C = 256, 4096, 192
M = B * HW
x = torch.randn(M, C, device="cuda").bfloat16()
y = torch.empty_like(x)
gam, bet = torch.randn(C, device="cuda"), torch.randn(C, device="cuda")
film = torch.randn(B, 2 * C, device="cuda")
mean, rstd = torch.empty(B * 32, device="cuda"), torch.empty(B * 32, device="cuda")
ws = ops.scratch_f32(torch.device("cuda", 0), lib().vaw_groupnorm_workspace_floats(B, HW, C))
for _ in range(3):
    assert lib().vaw_groupnorm_fwd(BF16, ptr(x), ptr(gam), ptr(bet), ptr(film), ptr(film) + 4 * C, 2 * C, 1, ptr(y), ptr(mean), ptr(rstd), B,
                                   HW, C, 32, 1e-5, ptr(ws), stream_ptr()) == 0
torch.cuda.synchronize()
n = 200
buf = (ctypes.c_ulonglong * n)()
lib().vaw_debug_gnc_prof.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib().vaw_debug_gnc_prof(buf, n) == 0
t0 = buf[0] & ((1 << 56) - 1)
prev = t0
for v in buf:
    slot, t = v >> 56, v & ((1 << 56) - 1)
    if slot == 0:
        break
    print(f"slot {slot:2d}  t={(t - t0) / 100:8.2f} us  +{(t - prev) / 100:6.2f}")
    prev = t
