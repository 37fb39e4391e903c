#!/bin/bash
# Round-5 evidence run (on the GPU box through gpurun).  Output: gpurun_out/r05_evidence/ (tools/make_r05_profiles.py copies
# what is to be judged into profiles/).
#   * the default bench line (twice: before and after the profile records it reads back are written on this box);
#   * rocprofv3 kernel stats of the SAME command shape (50 steps per launch, two agent groups, spin-up on) and of one launch
#     per pass, PMC passes (FETCH / WRITE / SQ) of the control kernel;
#   * the packed kernel at the short-horizon BASELINE shapes: rocprofv3 stats + SQ counters (tools/pack_point.py);
#   * phi_k streaming kernel: FETCH_SIZE of this round's build (tools/phik_pmc.sh);
#   * parity report, analytic checks, tick kernels / fleet tick (tools/r05_tick_profile.sh), two-rank evidence
#     (tools/r05_two_rank_probe.sh).
set -u
OUT=gpurun_out/r05_evidence
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
python3 bench.py --steps 20 --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.err"
bash tools/profile_r.sh r05_spl50 > /dev/null 2>&1
bash tools/profile_r.sh r05_spl1 --steps-per-launch 1 > /dev/null 2>&1
for g in spl50 spl1; do
  cp gpurun_out/prof_r05_$g/summary.txt "$OUT/${g}_summary.txt"; cp gpurun_out/prof_r05_$g/summary.json "$OUT/${g}_summary.json"
  cp gpurun_out/prof_r05_$g/trace/trace_kernel_stats.csv "$OUT/${g}_kernel_stats.csv" 2>/dev/null
done
# packed kernel: configs[1] at 8 lanes per agent x 24576 agents, yaml T = 50 at 16 lanes x 12288, configs[0] at 8 lanes x 32768
{
for cfg in "1 8 24576" "2 16 12288" "0 8 32768"; do
  set -- $cfg
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/pack_trace_s$1" -o t -- python3 tools/pack_point.py --shape $1 --lanes $2 --agents $3 --spl 50 --launches 20 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY --kernel-include-regex control_ --output-format csv -d "$OUT/pack_pmc_s$1" -o p -- python3 tools/pack_point.py --shape $1 --lanes $2 --agents $3 --spl 1 --launches 6 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE FETCH_SIZE --kernel-include-regex control_ --output-format csv -d "$OUT/pack_pmc2_s$1" -o p -- python3 tools/pack_point.py --shape $1 --lanes $2 --agents $3 --spl 1 --launches 6 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --kernel-include-regex control_ --output-format csv -d "$OUT/pack_pmc3_s$1" -o p -- python3 tools/pack_point.py --shape $1 --lanes $2 --agents $3 --spl 1 --launches 6 > /dev/null 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
names = {"0": "configs[0] (Omni K=5 T=5), 8 lanes per agent, 32768 agents", "1": "configs[1] (SimpleCart K=10 T=20), 8 lanes per agent, 24576 agents",
         "2": "explore_omni.yaml (K=10 T=50), 16 lanes per agent, 12288 agents"}
for s in ("1", "2", "0"):
    print("== control_pack_kernel, %s" % names[s])
    for p in glob.glob("%s/pack_trace_s%s/**/*kernel_stats.csv" % (out, s), recursive=True):
        for r in csv.DictReader(open(p)):
            if "control_pack" in r["Name"]:
                print("   rocprofv3 --kernel-trace --stats (50 steps per launch): calls=%s avg_ns=%s  -> %.2f us per pass" % (r["Calls"], r["AverageNs"], float(r["AverageNs"]) / 50e3))
    acc = collections.defaultdict(list)
    for sub in ("pack_pmc", "pack_pmc2", "pack_pmc3"):
        for p in glob.glob("%s/%s_s%s/**/*counter_collection.csv" % (out, sub, s), recursive=True):
            for r in csv.DictReader(open(p)):
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    m = {k: sum(v) / len(v) for k, v in acc.items()}
    w = m.get("SQ_WAVES", 1.0)
    print("   per launch (one step): " + "  ".join("%s %.4g" % (k, v) for k, v in sorted(m.items())))
    if "SQ_INSTS_VALU" in m:
        print("   per wavefront: VALU (incl. matrix) %.0f, matrix %.0f, LDS %.0f, SALU %.0f; SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES = %.3f; HBM read x2 %.2f MB, write %.2f MB per launch" %
              (m["SQ_INSTS_VALU"] / w, m.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0) / w, m.get("SQ_INSTS_LDS", 0) / w, m.get("SQ_INSTS_SALU", 0) / w,
               m.get("SQ_WAIT_INST_ANY", 0) / max(1.0, m.get("SQ_WAVE_CYCLES", 1)), 2 * m.get("FETCH_SIZE", 0) * 1024 / 1e6, m.get("WRITE_SIZE", 0) * 1024 / 1e6))
PY
} > "$OUT/pack_kernel_profile.txt" 2>&1
PHIK_CASES=16384:10:f64 bash tools/phik_pmc.sh > "$OUT/phik_pmc.txt" 2>&1
python3 tools/parity_report.py > "$OUT/parity_report.txt" 2>&1
EEA_PRINT_WORST=1 python3 -m pytest tests/test_analytic_checks.py -m gpu -q -s 2>&1 | grep -E "digits|closed-form|passed|failed" > "$OUT/analytic_checks.txt"
python3 tools/make_r05_profiles.py > /dev/null 2>&1
python3 bench.py --steps 20 --warmup 5 > "$OUT/bench_final.json" 2> "$OUT/bench_final.err"
