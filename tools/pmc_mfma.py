#!/usr/bin/env python3
"""Fold one rocprofv3 SQ/GRBM PMC pass into per-kernel MFMA utilisation and wave-cycle breakdown.

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \\
        --kernel-trace --output-format csv -d gpurun_out/pmc_m -o m -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-trace
    python tools/pmc_mfma.py gpurun_out/pmc_m/m_counter_collection.csv out.json

mfma_util_pct = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs) * 100: rocprofv3's MfmaUtil expression
(256 CUs x 4 SIMDs) with reduce(GRBM_GUI_ACTIVE, max) -- the CSV reports the counter SUMMED over the 8 XCDs (checked:
sum / 8 / kernel duration = 2.44 GHz), hence the division by 8.  The wait shares are fractions of SQ_WAVE_CYCLES: WAIT_ANY = parked on s_waitcnt / barrier,
WAIT_INST_ANY = issue stall, ACTIVE_INST_ANY = issuing (MI355X_MICROARCH.md, "rocprofv3 PMC slots")."""
import collections
import csv
import json
import sys


def main():
    path, out = sys.argv[1:3]
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.Counter()
    seen = set()
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        per[k][r["Counter_Name"]] += float(r["Counter_Value"])
        d = (k, r["Dispatch_Id"])
        if d not in seen:
            seen.add(d)
            calls[k] += 1
    rows = {}
    for k, c in per.items():
        act = c.get("GRBM_GUI_ACTIVE", 0.0)
        wc = c.get("SQ_WAVE_CYCLES", 0.0)
        if act <= 0:
            continue
        rows[k[:110]] = {"launches": calls[k], "gpu_active_cycles": act,
                         "mfma_util_pct": round(100.0 * c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (act / 8 * 1024), 2),
                         "wait_any_share": round(c.get("SQ_WAIT_ANY", 0.0) / wc, 3) if wc else None,
                         "wait_inst_share": round(c.get("SQ_WAIT_INST_ANY", 0.0) / wc, 3) if wc else None,
                         "active_inst_share": round(c.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, 3) if wc else None}
    top = dict(sorted(rows.items(), key=lambda kv: -kv[1]["gpu_active_cycles"])[:16])
    tot_act = sum(v["gpu_active_cycles"] for v in rows.values())
    tot_mfma = sum(v["mfma_util_pct"] * v["gpu_active_cycles"] for v in rows.values())
    rec = {"whole_run_mfma_util_pct": round(tot_mfma / tot_act, 2), "kernels": top,
           "note": "one PMC pass (kernels serialised by the profiler); mfma_util = MFMA-pipe busy cycles over all 1024 SIMDs"}
    json.dump(rec, open(out, "w"), indent=1)
    print(json.dumps({"whole_run_mfma_util_pct": rec["whole_run_mfma_util_pct"]}))
    for k, v in list(top.items())[:8]:
        print(f"{k[:64]:64s} mfma {v['mfma_util_pct']:6.2f}%  wait {v['wait_any_share']}  stall {v['wait_inst_share']}  issue {v['active_inst_share']}")


if __name__ == "__main__":
    main()
