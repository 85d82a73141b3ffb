#!/usr/bin/env python3
"""Which outputs of the fused LayerNorm-backward + gate-backward kernel differ from the pair of kernels, and where (GPU box)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vaw_amd  # noqa
from vaw_amd import ops
from vaw_amd._lib import lib, ptr, stream_ptr

B, T, D = (int(v) for v in sys.argv[1:4])
dtype = torch.bfloat16
g = torch.Generator().manual_seed(1)
r = lambda *s: torch.randn(*s, generator=g)
x = (r(B * T, D) * 2 + 0.5).cuda(); mod = (r(B, 6 * D) * 0.5).cuda(); dout = r(B * T, D).to(dtype).cuda(); dres = r(B * T, D).cuda(); y = r(B * T, D).to(dtype).cuda()
mean, rstd = torch.empty(B * T, device="cuda"), torch.empty(B * T, device="cuda")
out = torch.empty(B * T, D, device="cuda", dtype=dtype)
dt = ops.dt_of(out)
ops.ln_modulate_fwd(dt, ptr(x), ptr(mod) + 4 * 3 * D, ptr(mod) + 4 * 4 * D, 6 * D, ptr(out), ptr(mean), ptr(rstd), B, T, D)

def run(fused):
    dmod = torch.zeros(B, 6 * D, device="cuda")
    dx, dy, part = torch.empty(B * T, D, device="cuda"), torch.empty(B * T, D, device="cuda", dtype=dtype), torch.empty(B, D, device="cuda")
    a = (dt, ptr(dout), ptr(x), ptr(mean), ptr(rstd), ptr(mod) + 4 * 4 * D, 6 * D, ptr(dres), ptr(dx), ptr(dmod) + 4 * 3 * D, ptr(dmod) + 4 * 4 * D, 6 * D)
    if fused:
        ws = ops._row_ws(B, T, D)
        ops.check(lib().vaw_ln_modulate_bwd_gate(*a, ptr(y), ptr(mod) + 4 * 5 * D, ptr(dy), ptr(dmod) + 4 * 5 * D, ptr(part), B, T, D, ws.data_ptr(), ws.numel(), stream_ptr()), "x")
    else:
        ops.ln_modulate_bwd(*a, B, T, D)
        ops.gate_bwd(dt, ptr(dx), ptr(y), ptr(mod) + 4 * 5 * D, 6 * D, ptr(dy), ptr(dmod) + 4 * 5 * D, 6 * D, B, T, D, ptr(part))
    torch.cuda.synchronize()
    return dict(dx=dx, dy=dy, dshift=dmod[:, 3 * D:4 * D], dscale=dmod[:, 4 * D:5 * D], dgate=dmod[:, 5 * D:], part=part)

u, f = run(False), run(True)
f2 = run(True)
print('fused twice identical:', {k: bool(torch.equal(f[k], f2[k])) for k in f})
u2 = run(False)
print('pair twice identical:', {k: bool(torch.equal(u[k], u2[k])) for k in u})
for k in u:
    d = (u[k].double() - f[k].double()).abs()
    bad = (d > 0).nonzero()
    print(k, "max diff", float(d.max()), "n bad", len(bad), "first", bad[:4].tolist(), "rel", float((d / (u[k].double().abs() + 1e-30)).max()))
