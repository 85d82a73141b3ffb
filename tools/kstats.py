#!/usr/bin/env python3
"""Per-step view of a rocprofv3 *_kernel_stats.csv:  python tools/kstats.py <csv> <steps counted in the run> [top]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
tot = sum(float(r["TotalDurationNs"]) for r in rows)
calls = sum(int(r["Calls"]) for r in rows)
print(f"total {tot / 1e6 / steps:.3f} ms/step, {calls / steps:.0f} launches/step")
gemm = sum(float(r["TotalDurationNs"]) for r in rows if any(k in r["Name"] for k in ("gemm_", "splitk_reduce", "p8_group_fixup")))
print(f"GEMM family {gemm / 1e6 / steps:.3f} ms/step, everything else {(tot - gemm) / 1e6 / steps:.3f}")
for r in rows[:top]:
    print(f"{r['Name'][:96]:96s} {int(r['Calls']) / steps:7.1f}/step {float(r['TotalDurationNs']) / 1e6 / steps:8.3f} ms  avg {float(r['AverageNs']) / 1e3:8.1f} us")
