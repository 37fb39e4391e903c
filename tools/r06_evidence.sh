#!/bin/bash
# Round-6 evidence run (on the GPU box through gpurun).  Output: gpurun_out/r06_evidence/ (tools/make_r06_profiles.py copies what
# is to be judged into profiles/ and writes the records bench.py reads back).
#   * the default bench line + bench_detail.json (before and after the profile records are written on this box);
#   * rocprofv3 kernel stats of the SAME command shape (50 steps per launch, two agent groups, spin-up on) and of one launch per
#     pass, PMC passes (FETCH / WRITE / SQ) of the control kernel (tools/profile_r.sh);
#   * the short-horizon legs: rocprofv3 stats per leg in the leg's own launch form (tools/r06_pack_profile.sh), SQ counters
#     (tools/r06_pack_pmc.sh), the same-box table over batch sizes (tools/r06_pack_points.sh);
#   * the consensus exchange by protocol from the C++ host loop + host cost of the runtime calls (tools/r06_exchange_modes.sh);
#   * phi_k streaming kernel FETCH_SIZE (tools/phik_pmc.sh); parity report; analytic checks.
set -u
OUT=gpurun_out/r06_evidence
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
python3 bench.py --steps 20 --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.err"
bash tools/profile_r.sh r06_spl50 > /dev/null 2>&1
bash tools/profile_r.sh r06_spl1 --steps-per-launch 1 > /dev/null 2>&1
for g in spl50 spl1; do
  cp gpurun_out/prof_r06_$g/summary.txt "$OUT/${g}_summary.txt"; cp gpurun_out/prof_r06_$g/summary.json "$OUT/${g}_summary.json"
  cp gpurun_out/prof_r06_$g/trace/trace_kernel_stats.csv "$OUT/${g}_kernel_stats.csv" 2>/dev/null
done
bash tools/r06_pack_profile.sh r06ev > /dev/null 2>&1
cp gpurun_out/pack_prof_r06ev/summary.txt "$OUT/pack_summary.txt"; cp gpurun_out/pack_prof_r06ev/pack_profile.json "$OUT/pack_profile.json"
for i in 2 3; do cp gpurun_out/pack_prof_r06ev/c$i/trace_kernel_stats.csv "$OUT/pack_c${i}_kernel_stats.csv" 2>/dev/null; done
bash tools/r06_pack_pmc.sh r06ev > /dev/null 2>&1
cp gpurun_out/pack_pmc_r06ev/summary.txt "$OUT/pack_pmc.txt"
bash tools/r06_pack_points.sh > "$OUT/pack_points.txt" 2>&1
bash tools/r06_exchange_modes.sh > "$OUT/exchange_modes.txt" 2>&1
bash tools/r06_packed_consensus.sh > "$OUT/packed_consensus.txt" 2>&1
PHIK_CASES=16384:10:f64 bash tools/phik_pmc.sh > "$OUT/phik_pmc.txt" 2>&1
python3 tools/parity_report.py > "$OUT/parity_report.txt" 2>&1
EEA_PRINT_WORST=1 python3 -m pytest tests/test_analytic_checks.py -m gpu -q -s 2>&1 | grep -E "digits|closed-form|passed|failed" > "$OUT/analytic_checks.txt"
python3 tools/make_r06_profiles.py > /dev/null 2>&1
python3 bench.py --steps 20 --warmup 5 > "$OUT/bench_final.json" 2> "$OUT/bench_final.err"
cp bench_detail.json "$OUT/bench_detail.json"
python3 tools/make_r06_profiles.py > "$OUT/make_profiles.log" 2>&1
