#!/bin/bash
# Round 4, VERDICT r03 item 3: where SQ_LDS_BANK_CONFLICT of the control kernel comes from.  Same box: PMC pass of the
# shipping library and of a variant library (EEA_LIB_VARIANT), then interleaved timing rounds.
#   tools/r04_lds_ab.sh <outdir> <variant> [more variants]      (variants: suffixes of lib/libergodic_amd<suffix>.so)
OUT=${1:-gpurun_out/lds_ab}; shift
mkdir -p "$OUT"
export TMPDIR=/tmp
LEGS="--cpu-seconds 0 --no-latency --no-exchange --no-phik --no-grid-tile"
export EEA_BENCH_SPINUP_PASSES=0
for v in main "$@"; do
  lv=$v; [ "$v" = main ] && lv=""
  export EEA_LIB_VARIANT=$lv
  rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS \
    --kernel-include-regex control_ --output-format csv -d "$OUT/pmc_$v" -o pmc -- python3 bench.py --steps 2 --warmup 1 --passes-per-step 20 $LEGS > "$OUT/pmc_$v.log" 2>&1
  python3 - "$OUT/pmc_$v" "$v" <<'PY'
import csv, glob, sys, collections
d, v = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), vals in sorted(acc.items()):
    print("%-8s %-60s %-24s per dispatch %12.0f  (n=%d)" % (v, k, c, sum(vals) / len(vals), len(vals)))
PY
done
unset EEA_BENCH_SPINUP_PASSES EEA_LIB_VARIANT
ROUNDS=${ROUNDS:-3} tools/ab_variants.sh "main $*"
