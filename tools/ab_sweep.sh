#!/bin/bash
# Same-box A/B of two library builds over the BASELINE shapes: the shipping library against
# ergodic_exploration_amd/lib/libergodic_amd_old.so (copy the previous build there first).  Run through gpurun.
run() { label=$1; shift; for v in "" _old; do
  out=$(EEA_LIB_VARIANT=$v python3 bench.py --steps 10 --warmup 3 --passes-per-step 100 --cpu-seconds 0 --no-latency --no-exchange --no-phik "$@" 2>/dev/null | tail -1)
  echo "$label [variant '$v'] $(echo "$out" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.4g opt/s  %.1f us/pass" % (d["value"], 1e3*d["ms_per_pass"]))')"; done; }
run "yaml  omni  K10 T50  f64 n_mem=100" --model omni --num-basis 10 --horizon 5.0 --n-mem 100
run "cfg4  cart  K10 T200 f64" --model simple_cart --num-basis 10 --horizon 20.0
run "cfg4  cart  K10 T200 f64 n_mem=100" --model simple_cart --num-basis 10 --horizon 20.0 --n-mem 100
run "cfg4  cart  K10 T200 f32" --model simple_cart --num-basis 10 --horizon 20.0 --precision f32
run "K12 T200 f64 (generic)" --model simple_cart --num-basis 12 --horizon 20.0
run "K16 T200 f64 (generic)" --model simple_cart --num-basis 16 --horizon 20.0
run "cfg3  omni  K20 T250 f32" --model omni --num-basis 20 --horizon 5.0 --dt 0.02 --precision f32
run "cfg3  omni  K20 T250 f64" --model omni --num-basis 20 --horizon 5.0 --dt 0.02
