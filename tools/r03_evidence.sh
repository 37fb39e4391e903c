#!/bin/bash
# Round-3 evidence run (on the GPU box through gpurun): the default bench line, rocprofv3 kernel stats + PMC passes
# of the SAME command shape (spin-up on, two agent groups) and of one launch per pass, the shader clock under
# sustained load (phase stamps of the A/B library), phi_k counters at the bench's 16384^2 grid, parity report,
# exchange cost breakdown, the other BASELINE shapes.
# Output: gpurun_out/r03_evidence/  (tools/make_r03_profiles.py copies what is to be judged into profiles/)
set -u
OUT=gpurun_out/r03_evidence
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
python3 bench.py --steps 20 --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.err"
bash tools/profile_r.sh r03_g2 > /dev/null 2>&1
bash tools/profile_r.sh r03_g1 --agent-groups 1 > /dev/null 2>&1
for g in g2 g1; do
  cp gpurun_out/prof_r03_$g/summary.txt "$OUT/${g}_summary.txt"; cp gpurun_out/prof_r03_$g/summary.json "$OUT/${g}_summary.json"
  cp gpurun_out/prof_r03_$g/trace/trace_kernel_stats.csv "$OUT/${g}_kernel_stats.csv" 2>/dev/null
done
# shader clock while the kernel runs, after 1000 passes of load: cycles (s_memtime) / lifetime (s_memrealtime)
EEA_PHASE_WARMUP=1000 python3 tools/phase_timing.py 4096 > "$OUT/phase_timing.txt" 2>&1
EEA_PHASE_WARMUP=1000 python3 tools/phase_timing.py 2048 >> "$OUT/phase_timing.txt" 2>&1
PHIK_CASES="16384:10:f64" bash tools/phik_pmc.sh > "$OUT/phik_pmc.txt" 2>&1
python3 tools/parity_report.py > "$OUT/parity_report.txt" 2>&1
EEA_PRINT_WORST=1 python3 -m pytest tests/test_analytic_checks.py -m gpu -q -s 2>&1 | grep -E "digits|passed|failed" > "$OUT/analytic_checks.txt"
python3 tools/ck_cost.py > "$OUT/ck_cost.txt" 2>&1
bash tools/config_sweep.sh > "$OUT/config_sweep.txt" 2>&1
# the profile records bench.py reads back, written on the box, then the default command once more: its
# roofline.*_profiled fields then come from THIS box and build (profiles/ on the box is scratch; the files travel in $OUT)
python3 tools/make_r03_profiles.py > /dev/null 2>&1
python3 bench.py --steps 20 --warmup 5 > "$OUT/bench_final.json" 2> "$OUT/bench_final.err"
ls -la "$OUT"
