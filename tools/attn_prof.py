#!/usr/bin/env python3
"""Stage timeline of one workgroup of attn_fwd_big on the DiT-XL/2 shape (library built with -DATTN_PROF=1:
make exp XN=attnprof XF=-DATTN_PROF=1 XSRC=attention_bwd_big; VAW_HIP_LIB=.../libvaw_hip_attnprof.so).  wall_clock64 (100 MHz) by
lane 0 of workgroup (1, 777): 1 entry, 2 slice ready (wait + barrier passed), 3 S^T MFMAs issued, 4 softmax done, (next 2 = P V done),
5 loop done, 6 drained + barrier, 7 O staged, 8 O stored."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vaw_amd  # noqa: E402,F401
from vaw_amd import ops  # noqa: E402
from vaw_amd._lib import lib, ptr  # noqa: E402

B, H, T, hd = 128, 16, 256, 72
D = H * hd
qkv = (torch.randn(B * T, 3 * D, device="cuda") * 0.5).bfloat16()
o = torch.empty(B * T, D, device="cuda", dtype=torch.bfloat16)
lse = torch.empty(B * H * T, device="cuda")
desc = ops.attn_desc_token_major(B, H, T, hd)
for _ in range(3):
    ops.attn_fwd(ops.dt_of(o), desc, ptr(qkv), ptr(qkv) + 2 * D, ptr(qkv) + 4 * D, ptr(o), ptr(lse))
torch.cuda.synchronize()
n = 64
buf = (ctypes.c_ulonglong * n)()
lib().vaw_debug_attn_prof.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib().vaw_debug_attn_prof(buf, n) == 0
t0 = prev = buf[0] & ((1 << 56) - 1)
for v in buf:
    slot, t = v >> 56, v & ((1 << 56) - 1)
    if slot == 0:
        break
    print(f"slot {slot:2d}  t={(t - t0) / 100:8.2f} us  +{(t - prev) / 100:6.2f}")
    prev = t
