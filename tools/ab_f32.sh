#!/bin/bash
# Same-box A/B (shipping library against lib/libergodic_amd_old.so) of the fp32 shapes.  Run through gpurun.
run() { label=$1; shift; for v in "" _old; do
  out=$(EEA_LIB_VARIANT=$v python3 bench.py --steps 10 --warmup 3 --passes-per-step 100 --cpu-seconds 0 --no-latency --no-exchange --no-phik "$@" 2>/dev/null | tail -1)
  echo "$label [variant '$v'] $(echo "$out" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.4g opt/s  %.1f us/pass" % (d["value"], 1e3*d["ms_per_pass"]))')"; done; }
run "cfg4  cart  K10 T200 f32" --model simple_cart --num-basis 10 --horizon 20.0 --precision f32
run "cfg3  omni  K20 T250 f32" --model omni --num-basis 20 --horizon 5.0 --dt 0.02 --precision f32
run "cfg3  omni  K20 T250 f32 G1" --model omni --num-basis 20 --horizon 5.0 --dt 0.02 --precision f32 --agent-groups 1
