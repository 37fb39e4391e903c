#!/bin/bash
# per-kernel durations of the phi_k streaming kernel on large grids (rocprofv3 kernel trace of
# tools/phik_bench.py: every case launches spatial_pass1_kernel 7 times, in the order printed)
export TMPDIR=/tmp
OUT=gpurun_out/prof_phik
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 tools/phik_bench.py > $OUT/run.log 2>&1
python3 - <<'PY'
import csv, glob, os, statistics
cases = ["1024^2 K=30 f64", "4096^2 K=10 f64", "8192^2 K=10 f64", "8192^2 K=20 f64", "8192^2 K=30 f64",
         "8192^2 K=10 f32", "16384^2 K=10 f32",
         "1024^2 K=30 occ->f64", "8192^2 K=10 occ->f64", "16384^2 K=10 occ->f64", "32768^2 K=10 occ->f64",
         "32768^2 K=10 occ->f32", "16384^2 K=30 occ->f64"]
sizes = [1024**2*8, 4096**2*8, 8192**2*8, 8192**2*8, 8192**2*8, 8192**2*4, 16384**2*4,
         1024**2, 8192**2, 16384**2, 32768**2, 32768**2, 16384**2]
if os.environ.get("PHIK_CASES"):
    cases, sizes = [], []
    for x in os.environ["PHIK_CASES"].split(","):
        n, K, prec = x.split(":")
        cases.append("%s^2 K=%s %s" % (n, K, prec))
        sizes.append(int(n) ** 2 * {"f64": 8, "f32": 4, "occ64": 1, "occ32": 1}[prec])
rows = []
for p in glob.glob('gpurun_out/prof_phik/**/*kernel_trace.csv', recursive=True):
    rows += [r for r in csv.DictReader(open(p)) if ('spatial_pass1' in r['Kernel_Name'] or 'spatial_stream' in r['Kernel_Name'])]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
for i, name in enumerate(cases):
    chunk = rows[7 * i:7 * i + 7]
    if len(chunk) < 7:
        break
    us = statistics.median((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in chunk[2:])
    print("%-18s pass1 %8.1f us  %6.2f TB/s  grid %sx%s threads" % (name, us, sizes[i] / us / 1e6,
                                                                    chunk[0]['Grid_Size_X'], chunk[0]['Grid_Size_Y']))
PY
