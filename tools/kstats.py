#!/usr/bin/env python3
"""Per-step summary of a rocprofv3 --kernel-trace --stats kernel_stats.csv:  python tools/kstats.py <csv> <steps> [rows]"""
import csv
import re
import sys

path, steps = sys.argv[1], float(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 30
rows = list(csv.DictReader(open(path)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel ms/step {tot / steps / 1e6:.3f}")
for r in rows[:n]:
    name = re.sub(r"\(.*", "", r["Name"])[:64]
    print(f"{name:64s} calls/step {int(r['Calls']) / steps:7.1f}  ms/step {float(r['TotalDurationNs']) / steps / 1e6:7.3f}  avg us {float(r['AverageNs']) / 1e3:8.1f}")
