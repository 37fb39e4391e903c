#!/bin/bash
# Same-box A/B (shipping library against lib/libergodic_amd_old.so) of the K = 20 shapes.  Run through gpurun.
run() { label=$1; shift; for v in "" _old; do
  out=$(EEA_LIB_VARIANT=$v python3 bench.py --steps 10 --warmup 3 --passes-per-step 100 --cpu-seconds 0 --no-latency --no-exchange --no-phik "$@" 2>/dev/null | tail -1)
  echo "$label [variant '$v'] $(echo "$out" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.4g opt/s  %.1f us/pass" % (d["value"], 1e3*d["ms_per_pass"]))')"; done; }
run "cfg3  omni  K20 T250 f64" --model omni --num-basis 20 --horizon 5.0 --dt 0.02
run "cfg3  omni  K20 T250 f64 G1" --model omni --num-basis 20 --horizon 5.0 --dt 0.02 --agent-groups 1
run "K20 T200 cart f64" --model simple_cart --num-basis 20 --horizon 20.0
