#!/usr/bin/env python3
"""durations and start-to-start gaps of the record-sum kernel vs the control kernel in a rocprofv3 kernel trace
usage: tools/sum_trace.py <dir with *kernel_trace.csv>"""
import csv, glob, os, sys
import numpy as np
for p in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    rows = list(csv.DictReader(open(p, newline="")))
    for key in ("records_sum", "control_wave"):
        d = np.array([int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows if key in r["Kernel_Name"]], dtype=float)
        if len(d):
            print("%-14s n=%d  avg %.2f us  p50 %.2f  p90 %.2f  p99 %.2f  max %.2f" % (
                key, len(d), d.mean() / 1e3, np.percentile(d, 50) / 1e3, np.percentile(d, 90) / 1e3, np.percentile(d, 99) / 1e3, d.max() / 1e3))
    # the last 2000 sum launches (the lag-4 loop of ck_cost --quick): latency from the end of the later of the two
    # preceding control kernels to the end of the sum
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "sum" if "records_sum" in r["Kernel_Name"] else
                 ("ctl" if "control_wave" in r["Kernel_Name"] else "x")) for r in rows)
    last_ctl_end, lat = 0, []
    ends = []
    for s, e, k in ev:
        if k == "ctl":
            ends.append(e)
        elif k == "sum":
            prev = [x for x in ends[-6:] if x <= s]
            if prev:
                lat.append((s - max(prev), e - s))
    lat = np.array(lat[-2000:], dtype=float)
    if len(lat):
        print("last %d sums: wait after the preceding control kernel's end  avg %.2f us p90 %.2f;  duration avg %.2f us p90 %.2f" % (
            len(lat), lat[:, 0].mean() / 1e3, np.percentile(lat[:, 0], 90) / 1e3, lat[:, 1].mean() / 1e3, np.percentile(lat[:, 1], 90) / 1e3))
