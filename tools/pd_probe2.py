#!/usr/bin/env python3
"""Epilogue kinds of the parked-drain GEMM against the persistent kernel, with the location of mismatches (run on the GPU box).
    python tools/pd_probe2.py [tile] [Bt] [ragged]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vaw_amd  # noqa
from vaw_amd import ops
from vaw_amd._lib import lib, BF16, ptr

tile = int(sys.argv[1]) if len(sys.argv) > 1 else 10
Bt = int(sys.argv[2]) if len(sys.argv) > 2 else 130
ragged = int(sys.argv[3]) if len(sys.argv) > 3 else 24
T, D, H = 64, 768, 3072
M = Bt * T - ragged
g = torch.Generator().manual_seed(3)
x = torch.randn(M, D, generator=g).bfloat16().cuda()
w1 = (torch.randn(H, D, generator=g) * 0.05).bfloat16().cuda()
w2 = (torch.randn(D, H, generator=g) * 0.05).bfloat16().cuda()
b1, b2 = torch.randn(H, generator=g).cuda(), torch.randn(D, generator=g).cuda()
gate = torch.randn(Bt, D, generator=g).cuda()
resid = torch.randn(M, D, generator=g).cuda()
dy = torch.randn(M, D, generator=g).bfloat16().cuda()


def run(t):
    lib().vaw_debug_gemm_tile(t)
    out = {}
    a, hpre = ops.gemm_t(x, w1, bias=b1, act=1, want_aux=True)
    out["a"], out["hpre"] = a, hpre
    res = torch.full((M, D), 7777.0, device="cuda")
    aux = torch.full((M, D), 7777.0, device="cuda", dtype=torch.bfloat16)
    ops.gemm(BF16, 1, 1, M, D, H, ptr(a), H, ptr(w2), H, ptr(res), D, bias=ptr(b2), aux_out=ptr(aux), gate=ptr(gate), gate_ld=D,
             resid=ptr(resid), rows_per_batch=T, out_f32=True)
    out["res"], out["y"] = res, aux
    cs = torch.zeros(H, device="cuda")
    out["dh"] = ops.gemm_t(dy, w2, b_kmajor=False, act=2, aux_in=hpre, colsum_out=cs)
    out["dh_cs"] = cs
    torch.cuda.synchronize()
    lib().vaw_debug_gemm_tile(-1)
    return out


ref, got = run(3 if tile in (11, 14) else 2), run(tile)
rc = 0
for k in ("a", "hpre", "y", "res", "dh"):
    r, q = ref[k].float(), got[k].float()
    bad = (r != q) | (q != q)
    n = int(bad.sum())
    print(f"{k}: {n} of {bad.numel()} differ; nan in got {int((q != q).sum())}, untouched {int((q == 7777.0).sum())}")
    if n and k != "dh":
        rc = 1
        idx = bad.nonzero()
        rows = sorted(set(idx[:, 0].tolist()))
        print("   rows:", rows[:24], "... n_rows", len(rows), " rows%128:", sorted(set(v % 128 for v in rows))[:40])
        print("   cols/16:", sorted(set((idx[:, 1] // 16).tolist()))[:48])
        i0 = idx[0].tolist()
        print("   first:", i0, float(r[i0[0], i0[1]]), float(q[i0[0], i0[1]]))
    if k == "dh":
        rel = ((r - q).abs() / (r.abs() + 1e-20))
        print("   dh max rel diff", float(rel.max()), " > 2^-7:", int((rel > 2.0 ** -7).sum()))
        idx = (rel > 2.0 ** -6).nonzero()
        if len(idx):
            rows = sorted(set(idx[:, 0].tolist()))
            print("   bad rows:", rows[:24], "... n_rows", len(rows), " rows%128:", sorted(set(v % 128 for v in rows))[:48])
            print("   cols/16:", sorted(set((idx[:, 1] // 16).tolist()))[:64], "n", len(idx))
            i0 = idx[0].tolist()
            print("   first:", i0, float(r[i0[0], i0[1]]), float(q[i0[0], i0[1]]))
sys.exit(rc)
