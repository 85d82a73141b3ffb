#!/usr/bin/env python3
"""Per-kernel sums of every counter in a rocprofv3 counter_collection.csv, with a few ratios:  python tools/pmc_fold.py <csv> [name-filter]"""
import collections
import csv
import sys

per = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    per[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if (k, r["Dispatch_Id"]) not in seen:
        seen.add((k, r["Dispatch_Id"]))
        calls[k] += 1
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for k, c in per.items():
    if flt not in k:
        continue
    n = calls[k]
    wc = c.get("SQ_WAVE_CYCLES", 0.0)
    print(k[:90], f"launches {n}")
    for name, v in sorted(c.items()):
        extra = f"  ({v / wc:.3f} of wave cycles)" if wc and name.startswith("SQ_") and name != "SQ_WAVE_CYCLES" else ""
        print(f"    {name:34s} {v / n:16.0f} per launch{extra}")
