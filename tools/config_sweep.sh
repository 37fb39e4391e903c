#!/bin/bash
# throughput of the control kernels on the other BASELINE shapes (bench.py with explicit options), both dispatch
# paths: default (wavefront per agent where eligible) and --control-kernel workgroup (round 1's kernel)
run() { label=$1; shift; for path in auto workgroup; do
  out=$(python3 bench.py --control-kernel $path --steps 10 --warmup 3 --passes-per-step 100 --cpu-seconds 0 --no-latency --no-exchange --no-phik --no-grid-tile "$@" 2>/dev/null | tail -1)
  echo "$label [$path] $(echo "$out" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.4g opt/s  %.1f us/pass  agents %d x %d groups" % (d["value"], 1e3*d["ms_per_pass"], d["config"]["agents_per_gpu"], d["config"]["agent_groups"]))')"; done; }
run "cfg1  omni  K5  T5   f64" --model omni --num-basis 5 --horizon 0.5
run "cfg2  cart  K10 T20  f64" --model simple_cart --num-basis 10 --horizon 2.0
run "yaml  omni  K10 T50  f64 n_mem=100" --model omni --num-basis 10 --horizon 5.0 --n-mem 100
run "cfg4  cart  K10 T200 f64" --model simple_cart --num-basis 10 --horizon 20.0
run "cfg4  omni  K10 T200 f64" --model omni --num-basis 10 --horizon 20.0
run "cfg4  cart  K10 T200 f64 n_mem=100" --model simple_cart --num-basis 10 --horizon 20.0 --n-mem 100
run "cfg4  cart  K10 T200 f32" --model simple_cart --num-basis 10 --horizon 20.0 --precision f32
run "cfg3  omni  K20 T250 f32" --model omni --num-basis 20 --horizon 5.0 --dt 0.02 --precision f32
run "cfg3  omni  K20 T250 f64" --model omni --num-basis 20 --horizon 5.0 --dt 0.02
run "cfg5  omni  K30 T500 f64 (1024 agents)" --model omni --num-basis 30 --horizon 50.0 --agents 1024
