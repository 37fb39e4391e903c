#!/usr/bin/env python3
"""Gaps between consecutive control-kernel dispatches from a rocprofv3 --kernel-trace CSV: per queue (stream), the
time from one kernel's end to the next one's start, next to the kernel durations.  Usage: kernel_gaps.py <dir>"""
import csv
import glob
import sys
from collections import defaultdict


def main():
    files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
    rows = []
    for f in files:
        with open(f) as fh:
            rows += list(csv.DictReader(fh))
    per_q = defaultdict(list)
    for r in rows:
        if "control_wave_kernel" not in r["Kernel_Name"]:
            continue
        per_q[r["Queue_Id"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    for q, v in sorted(per_q.items()):
        v.sort()
        dur = [e - s for s, e in v]
        gap = [v[i + 1][0] - v[i][1] for i in range(len(v) - 1)]
        per = [v[i + 1][0] - v[i][0] for i in range(len(v) - 1)]
        gap.sort()
        dur.sort()
        per.sort()
        n = len(gap)
        print("queue %s: %d kernels; duration median %.2f us (p10 %.2f, p90 %.2f); end->start gap median %.2f us "
              "(p10 %.2f, p90 %.2f); start->start period median %.2f us" %
              (q, len(v), dur[len(dur) // 2] / 1e3, dur[len(dur) // 10] / 1e3, dur[9 * len(dur) // 10] / 1e3,
               gap[n // 2] / 1e3, gap[n // 10] / 1e3, gap[9 * n // 10] / 1e3, per[n // 2] / 1e3))


if __name__ == "__main__":
    main()
