#!/bin/bash
# Profiling recipe (run on the GPU box through gpurun): per-kernel stats + PMC passes for the
# headline bench.  Usage: tools/profile_r.sh <round-tag>   -> gpurun_out/prof_<tag>/
set -u
TAG=${1:-r02}
EXTRA_BENCH_ARGS=${EXTRA_BENCH_ARGS:-}
OUT=gpurun_out/prof_${TAG}
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python3 bench.py --steps 10 --warmup 2 --passes-per-step 5 --cpu-seconds 0 --no-latency --no-exchange --no-phik $EXTRA_BENCH_ARGS"
# 1) kernel trace + stats (durations)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- $BENCH > "$OUT/trace.log" 2>&1
# 2) PMC passes, each in its own run (FETCH_SIZE and WRITE_SIZE do not fit one pass)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex control_ --output-format csv -d "$OUT/pmc_fetch" -o pmc -- $BENCH > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --kernel-include-regex control_ --output-format csv -d "$OUT/pmc_write" -o pmc -- $BENCH > "$OUT/pmc_write.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY --kernel-include-regex control_ --output-format csv -d "$OUT/pmc_sq1" -o pmc -- $BENCH > "$OUT/pmc_sq1.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --kernel-include-regex control_ --output-format csv -d "$OUT/pmc_sq2" -o pmc -- $BENCH > "$OUT/pmc_sq2.log" 2>&1
find "$OUT" -name "*.csv" | head -50
python3 tools/summarize_prof.py "$OUT" "$TAG" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
