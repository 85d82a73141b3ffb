#!/usr/bin/env python3
"""Per-kernel means of every counter in a rocprofv3 --pmc counter_collection CSV (one row per kernel, one column per counter).
    python tools/pmc_table.py gpurun_out/pmc_x/x_counter_collection.csv [name-substring]"""
import collections
import csv
import re
import sys


def main():
    path = sys.argv[1]
    sub = sys.argv[2] if len(sys.argv) > 2 else ""
    val = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"])[:60]
        if sub and sub not in k:
            continue
        val[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
    names = sorted({c for v in val.values() for c in v})
    print("kernel".ljust(60), "n".rjust(4), *[c[-22:].rjust(22) for c in names])
    for k, v in sorted(val.items(), key=lambda kv: -sum(kv[1].values())):
        n = len(disp[k])
        print(k.ljust(60), str(n).rjust(4), *[f"{v.get(c, 0.0) / n:22.4g}" for c in names])


if __name__ == "__main__":
    main()
