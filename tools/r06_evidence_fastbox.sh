#!/bin/bash
# The boxes of the pool come in two speed classes (headline 0.357 - 0.365 and 0.378 - 0.383 of the fp64 vector peak: same
# kernel, same instruction counts).  This wrapper probes the box with a short headline run and runs tools/r06_evidence.sh only on
# a box of the faster class, so that the committed profiles are from ONE box and comparable with round 5's (0.377 - 0.381).
F=$(python3 bench.py --steps 6 --warmup 2 --no-grid-tile --no-exchange --no-latency --no-phik --no-other-configs --cpu-seconds 0 2>/dev/null | tail -1 | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['roofline']['frac'])")
echo "probe: headline frac $F"
if python3 -c "import sys; sys.exit(0 if float('$F') >= 0.370 else 1)"; then
  bash tools/r06_evidence.sh > gpurun_out/r06_evidence.log 2>&1
  echo "evidence done"
else
  echo "slow box: skipped"
fi
