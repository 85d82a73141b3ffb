#!/usr/bin/env python3
"""Fold two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE -- they do not fit one pass on gfx950) into HBM bytes per launch
of the GEMM kernel, with the corrections of /opt/skills/guides/MI355X_MICROARCH.md (HBM section): both counters are in
KiB; on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced streaming reads (16 B/lane global_load and
LDS-DMA alike), so it is doubled; WRITE_SIZE is exact for 16-B-per-lane stores.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_f -o f -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-trace
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_w -o w -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-trace
    python tools/pmc_traffic.py gpurun_out/pmc_f/f_counter_collection.csv gpurun_out/pmc_w/w_counter_collection.csv out.json [kernel-substring]
"""
import collections
import csv
import json
import sys


def load(path, counter):
    tot = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            t = tot[r["Kernel_Name"]]
            t[0] += 1
            t[1] += float(r["Counter_Value"])
    return tot


def main():
    fpath, wpath, out = sys.argv[1:4]
    key = sys.argv[4] if len(sys.argv) > 4 else "gemm_bf16_kernel"
    f, w = load(fpath, "FETCH_SIZE"), load(wpath, "WRITE_SIZE")
    rows = {}
    for k in f:
        n, fv = f[k]
        _, wv = w.get(k, [0, 0.0])
        rows[k] = {"launches": n, "fetch_bytes_per_launch": 2 * 1024 * fv / n, "write_bytes_per_launch": 1024 * wv / n}
    sel = {k: v for k, v in rows.items() if key in k}
    n = sum(v["launches"] for v in sel.values())
    total = sum(v["launches"] * (v["fetch_bytes_per_launch"] + v["write_bytes_per_launch"]) for v in sel.values())
    import datetime
    rec = {"kernel": key, "launches": n, "hbm_bytes_per_launch": total / n,
           "collected": datetime.datetime.now(datetime.timezone.utc).strftime("%Y-%m-%dT%H:%MZ"),
           "note": "FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, KiB -> bytes; separate PMC passes of the same command",
           "by_kernel": {k[:100]: v for k, v in sorted(sel.items(), key=lambda kv: -kv[1]["launches"])}}
    json.dump(rec, open(out, "w"), indent=1)
    print(json.dumps({k: rec[k] for k in ("kernel", "launches", "hbm_bytes_per_launch")}))


if __name__ == "__main__":
    main()
