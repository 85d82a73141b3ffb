#!/usr/bin/env python3
"""Fold a log of interleaved tools/gemm_bench.py runs ("== tag" lines between them) into one table: best time per shape and tag.
    python tools/ab_table.py LOG [base-tag]"""
import collections
import re
import sys

d = collections.defaultdict(lambda: collections.defaultdict(list))
tags, v = [], None
for l in open(sys.argv[1]):
    if l.startswith("=="):
        v = l.split()[1]
        if v not in tags:
            tags.append(v)
        continue
    m = re.match(r"(?:\[tile (\S+)\] )?(\S+ \S+)\s+\S+\s+M=.*?([\d.]+) us", l)
    if m and v:
        d[(m.group(2), m.group(1) or "")][v].append(float(m.group(3)))
base = sys.argv[2] if len(sys.argv) > 2 else tags[0]
for (k, tile), x in d.items():
    b = min(x[base]) if x.get(base) else None
    row = f"{k:12s} {tile:8s}"
    for t in tags:
        if x.get(t):
            row += f"  {t} {min(x[t]):7.1f}" + (f" ({100 * (min(x[t]) / b - 1):+5.1f}%)" if b and t != base else "")
    print(row)
