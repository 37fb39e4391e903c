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
    # the bench line the traced run itself printed (last JSON line of trace.log)
    bench = None
    try:
        with open(os.path.join(out, "trace.log")) as f:
            for line in f:
                if line.startswith("{") and '"metric"' in line:
                    bench = json.loads(line)
    except OSError:
        pass
    # kernel trace: register / LDS use of the control kernel, and the dispatches of the TIMED region
    for p in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
        disp = []
        for r in rows(p):
            if "control_" in r.get("Kernel_Name", ""):
                if "control_kernel_resources" not in summary:
                    keys = ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size",
                            "Workgroup_Size_X", "Grid_Size_X")
                    summary["control_kernel_resources"] = {k: r.get(k) for k in keys if k in r}
                    print("== control kernel resources:", summary["control_kernel_resources"])
                disp.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
        if bench is not None and disp:
            disp.sort()
            G = int(bench["config"]["agent_groups"])
            SPL = int(bench["config"].get("steps_per_launch", 1))   # receding-horizon steps (passes) per launch
            # the shard leg is followed by the short one-launch-per-pass leg (3 + 1 steps of 400 passes) when SPL > 1
            tail = (3 + 1) * 400 * G if (SPL > 1 and "single_launch_per_pass" in bench and
                                         bench["single_launch_per_pass"]["ms_per_pass"] != bench["ms_per_pass"]) else 0
            n_timed = int(bench["steps"]) * int(bench["config"]["passes_per_step"]) // SPL * G
            timed = disp[-(n_timed + tail):len(disp) - tail] if tail else disp[-n_timed:]
            dur = [e - s for s, e in timed]
            avg_ns = sum(dur) / len(dur)
            span_ns = max(e for _, e in timed) - min(s for s, _ in timed)
            passes = n_timed // G * SPL
            tr = {"dispatches_timed_region": len(timed), "dispatches_total": len(disp), "concurrent_launches": G,
                  "agents_per_launch": bench["roofline"]["agents_per_launch"],
                  "kernel_avg_us_timed_region": avg_ns * 1e-3,
                  "kernel_avg_us_all_dispatches": sum(e - s for s, e in disp) / len(disp) * 1e-3,
                  "pass_period_us_from_trace": span_ns * 1e-3 / passes,
                  "bench_ms_per_pass_same_run": bench["ms_per_pass"],
                  "bench_launch_ms_same_run": bench["roofline"]["launch_ms"],
                  "flops_per_launch": bench["roofline"]["flops_per_launch"]}
            # launches_per_pass x avg / concurrent launches = pass period the kernel durations alone would give
            tr["steps_per_launch"] = SPL
            tr["pass_us_from_kernel_avg"] = tr["kernel_avg_us_timed_region"] / SPL
            if tail:
                single = disp[-(3 * 400 * G):]   # the timed part of the one-launch-per-pass leg
                tr["single_launch_kernel_avg_us"] = sum(e - s for s, e in single) / len(single) * 1e-3
                tr["single_launch_pass_period_us_from_trace"] = (max(e for _, e in single) - min(s for s, _ in single)) * 1e-3 / (3 * 400)
            tr["tflops_from_kernel_avg"] = G * tr["flops_per_launch"] / (avg_ns * 1e-9) / 1e12
            tr["frac_of_78.6_from_kernel_avg"] = tr["tflops_from_kernel_avg"] / 78.6
            tr["frac_same_run_bench_line"] = bench["roofline"]["frac"]
            summary["timed_region"] = tr
            print("== timed region (last %d of %d control dispatches; %d concurrent launches per pass)" % (
                len(timed), len(disp), G))
            for k, v in tr.items():
                print("   %-34s %s" % (k, ("%.6g" % v) if isinstance(v, float) else v))
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
