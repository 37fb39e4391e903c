#!/bin/bash
# Round-2 evidence run (on the GPU box through gpurun): bench line, rocprofv3 kernel stats + PMC passes for the
# default command (two agent groups) and for one launch per pass, phase timing, the other BASELINE shapes.
# Output: gpurun_out/r02_evidence/  (copy what is to be judged into profiles/)
set -u
OUT=gpurun_out/r02_evidence
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
bash tools/profile_r.sh r02_g2 > /dev/null 2>&1
EXTRA_BENCH_ARGS="--agent-groups 1" bash tools/profile_r.sh r02_g1 > /dev/null 2>&1
cp gpurun_out/prof_r02_g2/summary.txt "$OUT/g2_summary.txt"; cp gpurun_out/prof_r02_g2/summary.json "$OUT/g2_summary.json"
cp gpurun_out/prof_r02_g1/summary.txt "$OUT/g1_summary.txt"; cp gpurun_out/prof_r02_g1/summary.json "$OUT/g1_summary.json"
cp gpurun_out/prof_r02_g2/trace/trace_kernel_stats.csv "$OUT/g2_kernel_stats.csv" 2>/dev/null
cp gpurun_out/prof_r02_g1/trace/trace_kernel_stats.csv "$OUT/g1_kernel_stats.csv" 2>/dev/null
python3 tools/phase_timing.py 4096 > "$OUT/phase_timing.txt" 2>&1
python3 tools/phase_timing.py 1024 >> "$OUT/phase_timing.txt" 2>&1
EEA_CONTROL_PATH=workgroup python3 tools/phase_timing.py 4096 >> "$OUT/phase_timing.txt" 2>&1
bash tools/config_sweep.sh > "$OUT/config_sweep.txt" 2>&1
python3 tools/parity_report.py > "$OUT/parity_report.txt" 2>&1
python3 tools/rebuild_bench.py > "$OUT/rebuild.txt" 2>&1
ls -la "$OUT"
