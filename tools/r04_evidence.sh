#!/bin/bash
# Round-4 evidence run (on the GPU box through gpurun).  Output: gpurun_out/r04_evidence/ (tools/make_r04_profiles.py
# copies what is to be judged into profiles/).
#   * the default bench line (twice: before and after the profile records it reads back are written on this box);
#   * rocprofv3 kernel stats of the SAME command shape (50 steps per launch, two agent groups, spin-up on) and of one launch
#     per pass, PMC passes (FETCH / WRITE / SQ) of the control kernel;
#   * BASELINE configs[2] (Omni, K = 20, T = 250, fp32): kernel stats + PMC of its bench shape;
#   * parity report (incl. the timed instances and the consensus leg's exact form), analytic checks;
#   * exchange cost ladder (local communicator and one-rank RCCL), rebuild kernel trace, phase timing / shader clock.
set -u
OUT=gpurun_out/r04_evidence
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
python3 bench.py --steps 20 --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.err"
bash tools/profile_r.sh r04_spl50 > /dev/null 2>&1
bash tools/profile_r.sh r04_spl1 --steps-per-launch 1 > /dev/null 2>&1
for g in spl50 spl1; do
  cp gpurun_out/prof_r04_$g/summary.txt "$OUT/${g}_summary.txt"; cp gpurun_out/prof_r04_$g/summary.json "$OUT/${g}_summary.json"
  cp gpurun_out/prof_r04_$g/trace/trace_kernel_stats.csv "$OUT/${g}_kernel_stats.csv" 2>/dev/null
done
# configs[2]: Omni K = 20 T = 250 fp32, the other_configs leg's shape
bash tools/profile_r.sh r04_k20_f32 --model omni --num-basis 20 --horizon 5.0 --dt 0.02 --precision f32 > /dev/null 2>&1
cp gpurun_out/prof_r04_k20_f32/summary.txt "$OUT/k20_f32_summary.txt"; cp gpurun_out/prof_r04_k20_f32/summary.json "$OUT/k20_f32_summary.json"
EEA_PHASE_WARMUP=1000 python3 tools/phase_timing.py 4096 > "$OUT/phase_timing.txt" 2>&1
EEA_PHASE_WARMUP=1000 python3 tools/phase_timing.py 2048 >> "$OUT/phase_timing.txt" 2>&1
python3 tools/parity_report.py > "$OUT/parity_report.txt" 2>&1
EEA_PRINT_WORST=1 python3 -m pytest tests/test_analytic_checks.py -m gpu -q -s 2>&1 | grep -E "digits|closed-form|passed|failed" > "$OUT/analytic_checks.txt"
python3 tools/exchange_cost.py > "$OUT/exchange_cost.txt" 2>&1
python3 tools/exchange_cost.py --rccl >> "$OUT/exchange_cost.txt" 2>&1
for impl in 0 1; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/rebuild_trace_$impl" -o t -- python3 tools/rebuild_trace.py $impl > /dev/null 2>&1
done
python3 - "$OUT" > "$OUT/rebuild_kernels.txt" <<'PY'
import csv, glob, sys
out = sys.argv[1]
print("# configTarget rebuild at the three BASELINE grids (121x61 K=10, 256x256 K=20, 1024x1024 K=30), Gaussian target: kernel")
print("# durations from rocprofv3 --kernel-trace, 50 enqueue-only rebuilds per grid (tools/rebuild_trace.py); stream time =")
print("# first start to last end of the grid's dispatches / 50 (includes the host's enqueue gaps)")
for impl, name, per in ((0, "per-axis factors (default): ONE launch", (1, 1, 1)),
                        (1, "Target::fill + streaming spatialCoeff (EEA_OPT_REBUILD_IMPL = 1): 2 / 3 / 3 launches", (2, 3, 3))):
    for f in glob.glob("%s/rebuild_trace_%d/**/*kernel_trace.csv" % (out, impl), recursive=True):
        rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
        rows = [r for r in rows if any(k in r["Kernel_Name"] for k in ("gaussian_phik", "target_fill", "spatial_stream", "sum_partials"))]
        print("== %s: %d kernel dispatches" % (name, len(rows)))
        pos = 0
        for grid, n in zip(("121x61", "256x256", "1024x1024"), per):
            part = rows[pos:pos + 50 * n]
            pos += 50 * n
            if not part:
                continue
            by = {}
            for r in part:
                by.setdefault(r["Kernel_Name"].split("<")[0].split("::")[-1], []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            span = (int(part[-1]["End_Timestamp"]) - int(part[0]["Start_Timestamp"])) / 50.0
            ksum = sum(sum(v) for v in by.values()) / 50.0
            print("   %-10s kernels per rebuild %6.2f us (stream time %6.2f us):  " % (grid, ksum * 1e-3, span * 1e-3) +
                  ";  ".join("%s avg %.2f us" % (k, sum(v) / len(v) * 1e-3) for k, v in by.items()))
PY
python3 tools/make_r04_profiles.py > /dev/null 2>&1
python3 bench.py --steps 20 --warmup 5 > "$OUT/bench_final.json" 2> "$OUT/bench_final.err"
ls -la "$OUT"
