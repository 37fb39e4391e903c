#!/bin/bash
# A/B of library variants / kernel versions on the headline bench (interleaved rounds, one process each).
# (EEA_CONTROL_IMPL=v1 and EEA_PHIK_IMPL=valu need the A/B library: make -C ergodic_exploration_amd/csrc AB=1, EEA_LIB_VARIANT=_ab)
# usage: tools/ab_bench.sh "<label>:<env assignments>" ...   e.g.  "v1:EEA_CONTROL_IMPL=v1" "w4:EEA_LIB_VARIANT=_w4" "main:"
ROUNDS=${ROUNDS:-3}
for r in $(seq 1 $ROUNDS); do
  for spec in "$@"; do
    label=${spec%%:*}; envs=${spec#*:}
    out=$(env $envs python3 bench.py --steps 50 --warmup 10 --cpu-seconds 0 2>/dev/null | tail -1)
    echo "$label $(echo "$out" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.4g opt/s  launch %.2f us" % (d["value"], 1e3*d["roofline"]["launch_ms"]))')"
  done
done
