#!/usr/bin/env python3
"""vaw_fp8_transpose and vaw_fp8_quantize_delayed (bf16 -> fp8 + transposed copy) at the DiT-XL/2 sizes, over rotating buffer sets
larger than the Infinity Cache.  Bytes: transpose = 2 R C; quantise = 2 R C (bf16 in) + 2 R C (q and qt out)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vaw_amd  # noqa: E402,F401
from vaw_amd._lib import BF16, BF8, FP8, lib, ptr, stream_ptr  # noqa: E402

NSET = 6
for (R, C) in [(32768, 1152), (32768, 4608), (32768, 3456)]:
    qs = [torch.randint(0, 255, (R, C), device="cuda", dtype=torch.uint8) for _ in range(NSET)]
    ts = [torch.empty(C, R, device="cuda", dtype=torch.uint8) for _ in range(NSET)]
    xs = [torch.randn(R, C, device="cuda").bfloat16() for _ in range(NSET)]
    state = torch.tensor([1.0, 0.0, 448.0, 0.0], device="cuda")

    def tr(i):
        assert lib().vaw_fp8_transpose(ptr(qs[i]), R, C, C, ptr(ts[i]), R, stream_ptr()) == 0

    def qd(i):
        assert lib().vaw_fp8_quantize_delayed(BF16, FP8, ptr(xs[i]), R, C, C, ptr(qs[i]), C, ptr(ts[i]), R, ptr(state), stream_ptr()) == 0

    for fn, nm, nbytes in ((tr, "transpose", 2 * R * C), (qd, "quantise+transpose", 4 * R * C)):
        for i in range(NSET):
            fn(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for it in range(4 * NSET):
            fn(it % NSET)
        e1.record()
        torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / (4 * NSET)
        print(f"R={R} C={C:5d} {nm:20s} {us:8.1f} us  {nbytes / us / 1e6:6.2f} TB/s", flush=True)
