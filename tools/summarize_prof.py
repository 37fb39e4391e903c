#!/usr/bin/env python3
"""Condenses the rocprofv3 CSV output of tools/profile_r.sh into a small text + JSON summary
(copied to profiles/ by hand after a gpurun call)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def rows(path):
    with open(path, newline="") as f:
        return list(csv.DictReader(f))


def main():
    out, tag = sys.argv[1], sys.argv[2]
    summary = {"tag": tag}
    # kernel stats
    for p in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
        print("== kernel stats (%s)" % os.path.relpath(p, out))
        for r in rows(p):
            name = r.get("Name", "")
            print("%-90s calls=%s total_ns=%s avg_ns=%s pct=%s" % (name[:90], r.get("Calls"), r.get("TotalDurationNs"),
                                                                   r.get("AverageNs"), r.get("Percentage")))
            if "control_kernel" in name or "control_wave_kernel" in name:
                summary["control_kernel_avg_ns"] = float(r.get("AverageNs", 0))
                summary["control_kernel_calls"] = int(r.get("Calls", 0))
    # kernel trace: register / LDS use of the control kernel
    for p in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
        for r in rows(p):
            if "control_" in r.get("Kernel_Name", ""):
                keys = ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size",
                        "Workgroup_Size", "Grid_Size")
                summary["control_kernel_resources"] = {k: r.get(k) for k in keys if k in r}
                print("== control kernel resources:", summary["control_kernel_resources"])
                break
    # counters
    counters = defaultdict(list)
    for p in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        for r in rows(p):
            if "control_" not in r.get("Kernel_Name", ""):
                continue
            counters[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("== PMC (control_kernel, mean per dispatch over %d dispatches)" % (max([len(v) for v in counters.values()] or [0])))
    pmc = {}
    for k in sorted(counters):
        v = counters[k]
        pmc[k] = sum(v) / len(v)
        print("%-32s %.6g" % (k, pmc[k]))
    summary["pmc_mean_per_dispatch"] = pmc
    if "FETCH_SIZE" in pmc or "WRITE_SIZE" in pmc:
        # FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE under-reports wide coalesced
        # streaming reads by 2x (MI355X_MICROARCH.md "HBM"): report both raw and corrected
        f_raw = pmc.get("FETCH_SIZE", 0.0) * 1024.0
        w_raw = pmc.get("WRITE_SIZE", 0.0) * 1024.0
        summary["hbm_read_bytes_raw"] = f_raw
        summary["hbm_read_bytes_x2_corrected"] = 2.0 * f_raw
        summary["hbm_write_bytes_raw"] = w_raw
        summary["hbm_bytes_per_launch"] = 2.0 * f_raw + w_raw
        print("HBM bytes per launch: read raw %.4g (x2 corrected %.4g), write %.4g" % (f_raw, 2 * f_raw, w_raw))
    with open(os.path.join(out, "summary.json"), "w") as f:
        json.dump(summary, f, indent=1)


if __name__ == "__main__":
    main()
