#!/bin/bash
# Profiling recipe (run on the GPU box through gpurun): rocprofv3 kernel trace + stats of the DEFAULT bench command
# shape (clock spin-up on, two agent groups, long passes-per-step) and separate PMC passes for the control kernel.
# Usage: tools/profile_r.sh <tag> [extra bench args]   -> gpurun_out/prof_<tag>/
set -u
TAG=${1:-r03}; shift || true
OUT=gpurun_out/prof_${TAG}
mkdir -p "$OUT"
export TMPDIR=/tmp
LEGS="--cpu-seconds 0 --no-latency --no-exchange --no-phik --no-grid-tile --no-other-configs"
# 1) kernel trace + stats on the default shape: 1000 spin-up passes + 1 warm-up step + 3 timed steps of 1000 passes.
#    (steps per launch as the default: 50; pass --steps-per-launch 1 for the one-launch-per-pass shape)
#    tools/summarize_prof.py takes the average over the dispatches of the TIMED region only (the last steps x passes x
#    groups control dispatches) and compares it with the ms_per_pass the same run printed
TRACE_BENCH="python3 bench.py --steps 3 --warmup 1 --passes-per-step 2000 $LEGS $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- $TRACE_BENCH > "$OUT/trace.log" 2>&1
# 2) PMC passes, each in its own run (FETCH_SIZE and WRITE_SIZE do not fit one pass); counter collection serialises
#    the dispatches, so these runs are short and skip the spin-up
export EEA_BENCH_SPINUP_PASSES=0
PMC_BENCH="python3 bench.py --steps 2 --warmup 1 --passes-per-step 100 --no-single-launch $LEGS $*"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex control_ --output-format csv -d "$OUT/pmc_fetch" -o pmc -- $PMC_BENCH > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --kernel-include-regex control_ --output-format csv -d "$OUT/pmc_write" -o pmc -- $PMC_BENCH > "$OUT/pmc_write.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY --kernel-include-regex control_ --output-format csv -d "$OUT/pmc_sq1" -o pmc -- $PMC_BENCH > "$OUT/pmc_sq1.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --kernel-include-regex control_ --output-format csv -d "$OUT/pmc_sq2" -o pmc -- $PMC_BENCH > "$OUT/pmc_sq2.log" 2>&1
unset EEA_BENCH_SPINUP_PASSES
python3 tools/summarize_prof.py "$OUT" "$TAG" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
