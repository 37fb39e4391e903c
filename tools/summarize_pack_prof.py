#!/usr/bin/env python3
"""tools/r06_pack_profile.sh's output -> summary.txt (stdout) + pack_profile.json: per leg the rocprofv3 kernel average over
the dispatches of the timed region (the last `launches_timed` control dispatches of the trace), the same over ALL dispatches
(what `--stats` prints: includes the cold probe and the ramp), and the event-timed pass the run itself printed."""
import csv
import glob
import json
import os
import sys


def main():
    out = sys.argv[1]
    cases = []
    for log in sorted(glob.glob(os.path.join(out, "c*.log"))):
        rec = None
        with open(log) as f:
            for line in f:
                if line.startswith("{") and '"launches_timed"' in line:
                    rec = json.loads(line)
        if rec is None:
            print("!! no record in", log)
            continue
        d = log[:-4]
        disp, res = [], None
        for p in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            with open(p, newline="") as f:
                for r in csv.DictReader(f):
                    if "control_" in r.get("Kernel_Name", ""):
                        disp.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
                        if res is None:
                            res = {k: r.get(k) for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size",
                                                         "Scratch_Size", "Workgroup_Size_X", "Grid_Size_X")}
        if not disp:
            print("!! no control dispatches for", rec["config"])
            continue
        disp.sort()
        n = rec["launches_timed"]
        timed = disp[-n:]
        spl, B = rec["steps_per_launch"], rec["agents"]
        avg = sum(e - s for s, e, _ in timed) / len(timed) * 1e-3
        avg_all = sum(e - s for s, e, _ in disp) / len(disp) * 1e-3
        span = (max(e for _, e, _ in timed) - min(s for s, _, _ in timed)) * 1e-3
        period = span / (n // 2 * spl)          # two concurrent launches per spl passes
        flops = rec["roofline"]["flops_per_optimisation"]
        peak = rec["roofline"]["peak"]
        frac_prof = flops * B / (avg / spl * 1e-6) / 1e12 / peak
        c = {"config": rec["config"], "agents": B, "lanes_per_agent": rec["lanes_per_agent"], "K": rec["num_basis"],
             "T": rec["horizon_steps"], "steps_per_launch": spl, "kernel": timed[-1][2].split("(")[0][:120], "resources": res,
             "dispatches_total": len(disp), "dispatches_timed_region": len(timed),
             "kernel_avg_us_timed_region": avg, "kernel_avg_us_all_dispatches": avg_all,
             "pass_us_from_kernel_avg": avg / spl, "pass_period_us_from_trace": period,
             "bench_us_per_pass_same_run": 1e3 * rec["ms_per_pass"],
             "rocprof_over_bench": (avg / spl) / (1e3 * rec["ms_per_pass"]),
             "frac_from_kernel_avg": frac_prof, "frac_same_run_events": rec["roofline"]["frac"]}
        cases.append(c)
        print("== %s  (%d agents, %d lanes per agent, K=%d T=%d, %d steps per launch)" % (c["config"], B, c["lanes_per_agent"],
                                                                                        c["K"], c["T"], spl))
        for k in ("kernel", "resources", "dispatches_total", "dispatches_timed_region", "kernel_avg_us_timed_region",
                  "kernel_avg_us_all_dispatches", "pass_us_from_kernel_avg", "pass_period_us_from_trace",
                  "bench_us_per_pass_same_run", "rocprof_over_bench", "frac_from_kernel_avg", "frac_same_run_events"):
            v = c[k]
            print("   %-32s %s" % (k, ("%.5g" % v) if isinstance(v, float) else v))
    with open(os.path.join(out, "pack_profile.json"), "w") as f:
        json.dump({"source": "tools/r06_pack_profile.sh (rocprofv3 --kernel-trace --stats, one stand-alone run per leg)",
                   "cases": cases}, f, indent=1)


if __name__ == "__main__":
    main()
