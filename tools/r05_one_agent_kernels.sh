#!/bin/bash
# Device time of ONE agent's control() by kernel (rocprofv3 --kernel-trace): the body a resident workgroup would run.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/one_agent; mkdir -p $OUT
run() {  # tag, args...
  tag=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag -o t -- python3 $R/tools/pack_point.py --agents 1 --launches 200 "$@" > $OUT/$tag.log 2>&1
  f=$(ls $OUT/$tag/*kernel_stats.csv 2>/dev/null | head -1)
  echo "== $tag: $*"; python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "control" in r["Name"]:
        print("   %-90s calls %s avg %.2f us min %.2f us" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
}
run c0_wave --shape 0 --lanes 64
run c0_pack8 --shape 0 --lanes 8
run c0_wg --shape 0 --lanes 64 --workgroup
run c1_wave --shape 1 --lanes 64
run c1_pack32 --shape 1 --lanes 32
run c1_wg --shape 1 --lanes 64 --workgroup
run y_wave --shape 2 --lanes 64
run y_wg --shape 2 --lanes 64 --workgroup
run c3_wave --shape 1 --lanes 64 --steps 200
run c3_wg --shape 1 --lanes 64 --steps 200 --workgroup
