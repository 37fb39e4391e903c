#!/bin/bash
# the short-horizon legs of bench.py's other_configs under rocprofv3 --kernel-trace --stats, each in the
# leg's own launch form (two agent groups x 50 steps per launch) behind a 1 s clock spin-up; the kernel average is taken over
# the dispatches of the TIMED region only (tools/summarize_pack_prof.py) and compared with the event-timed pass of the same run.
# Run on the GPU box:  tools/r06_pack_profile.sh [tag]   -> gpurun_out/pack_prof_<tag>/ (+ summary.txt / pack_profile.json)
set -u
TAG=${1:-r06}
OUT=gpurun_out/pack_prof_${TAG}
mkdir -p "$OUT"
export TMPDIR=/tmp
i=0
while IFS= read -r CASE; do
  i=$((i + 1))
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/c$i" -o trace -- python3 tools/other_config_point.py --case "$CASE" > "$OUT/c$i.log" 2>&1
done <<'CASES'
configs[0], chip-filling batch
configs[1], chip-filling batch
explore_omni.yaml as shipped, chip-filling batch
configs[0]
configs[1]
explore_omni.yaml as shipped (K = 10, T = 50)
CASES
python3 tools/summarize_pack_prof.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
