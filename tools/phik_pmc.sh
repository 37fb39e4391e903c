#!/bin/bash
# HBM traffic and matrix-pipe counters of the phi_k streaming kernel (separate --pmc passes, as for
# the control kernel): FETCH_SIZE (KiB, x2 on gfx950 for streamed reads), MFMA busy cycles, GRBM cycles
export TMPDIR=/tmp
export PHIK_CASES=${PHIK_CASES:-"8192:10:f64,8192:30:f64,16384:10:f32,16384:10:occ64"}
OUT=gpurun_out/pmc_phik
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex spatial_stream --output-format csv -d $OUT/fetch -o p -- python3 tools/phik_bench.py > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVES --kernel-include-regex spatial_stream --output-format csv -d $OUT/sq -o p -- python3 tools/phik_bench.py > $OUT/sq.log 2>&1
python3 - <<'PY'
import csv, glob, os
from collections import defaultdict
cases = os.environ["PHIK_CASES"].split(",")
for sub in ("fetch", "sq"):
    rows = []
    for p in glob.glob('gpurun_out/pmc_phik/%s/**/*counter_collection.csv' % sub, recursive=True):
        rows += list(csv.DictReader(open(p)))
    # dispatches in launch order; 7 launches per case
    by_disp = defaultdict(dict)
    for r in rows:
        by_disp[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(by_disp)
    for i, c in enumerate(cases):
        chunk = ids[7 * i:7 * i + 7]
        if len(chunk) < 7:
            break
        n, K, prec = c.split(":")
        nbytes = int(n) ** 2 * {"f64": 8, "f32": 4, "occ64": 1, "occ32": 1}[prec]
        vals = defaultdict(list)
        for d in chunk[2:]:
            for k, v in by_disp[d].items():
                vals[k].append(v)
        line = "%-18s" % c
        for k in sorted(vals):
            m = sum(vals[k]) / len(vals[k])
            if k == "FETCH_SIZE":
                line += "  FETCH_SIZE %.0f KiB raw -> x2 = %.1f MB (algorithmic %.1f MB, ratio %.2f)" % (
                    m, 2 * m * 1024 / 1e6, nbytes / 1e6, 2 * m * 1024 / nbytes)
            else:
                line += "  %s %.4g" % (k, m)
        print(line)
PY
