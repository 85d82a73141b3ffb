#!/usr/bin/env python3
"""Stage timeline of one workgroup of attn_bwd_t64_mfma on the DiT-B/4 shape (library built with -DATTN64_PROF=1:
make exp XN=a64prof XF=-DATTN64_PROF=1 XSRC=attention_mfma).  1 entry, 2 loads issued + delta computed, 3 images landed, 4 phase 1
(dQ) done, 5 phase 2 (dK, dV) done, 6 staged, 7 stored."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vaw_amd  # noqa: E402,F401
from vaw_amd import ops  # noqa: E402
from vaw_amd._lib import lib, ptr  # noqa: E402

B, H, T, hd = 256, 12, 64, 64
D = H * hd
qkv = (torch.randn(B * T, 3 * D, device="cuda") * 0.5).bfloat16()
do = torch.randn(B * T, D, device="cuda").bfloat16()
o = torch.empty(B * T, D, device="cuda", dtype=torch.bfloat16)
lse, delta = torch.empty(B * H * T, device="cuda"), torch.empty(B * H * T, device="cuda")
dqkv = torch.empty_like(qkv)
desc = ops.attn_desc_token_major(B, H, T, hd)
dt = ops.dt_of(o)
ops.attn_fwd(dt, desc, ptr(qkv), ptr(qkv) + 2 * D, ptr(qkv) + 4 * D, ptr(o), ptr(lse))
for _ in range(3):
    ops.attn_bwd(dt, desc, ptr(qkv), ptr(qkv) + 2 * D, ptr(qkv) + 4 * D, ptr(o), ptr(do), ptr(lse), ptr(delta), ptr(dqkv), ptr(dqkv) + 2 * D,
                 ptr(dqkv) + 4 * D)
torch.cuda.synchronize()
n = 16
buf = (ctypes.c_ulonglong * n)()
lib().vaw_debug_attn64_prof.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib().vaw_debug_attn64_prof(buf, n) == 0
t0 = prev = buf[0] & ((1 << 56) - 1)
for v in buf:
    slot, t = v >> 56, v & ((1 << 56) - 1)
    if slot == 0:
        break
    print(f"slot {slot:2d}  t={(t - t0) / 100:8.2f} us  +{(t - prev) / 100:6.2f}")
    prev = t
