#!/bin/bash
# threads per agent of the workgroup-per-agent kernel (--workgroup-threads) on the headline shape and the short-horizon shapes
run() { label=$1; blk=$2; shift 2; out=$(python3 bench.py --control-kernel workgroup --workgroup-threads $blk --steps 30 --warmup 5 --passes-per-step 100 --cpu-seconds 0 --no-exchange --no-phik --no-grid-tile "$@" 2>/dev/null | tail -1); echo "$label block=$blk $(echo "$out" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.4g opt/s  %.1f us/launch" % (d["value"], 1e3*d["roofline"]["launch_ms"]))')"; }
for b in 64 128 256; do
run "cfg1  omni  K5  T5   f64" $b --model omni --num-basis 5 --horizon 0.5
run "cfg2  cart  K10 T20  f64" $b --model simple_cart --num-basis 10 --horizon 2.0
run "yaml  omni  K10 T50  f64 n_mem=100" $b --model omni --num-basis 10 --horizon 5.0 --n-mem 100
run "yaml  cart  K10 T50  f64 n_mem=0" $b --model simple_cart --num-basis 10 --horizon 5.0
run "cfg4  cart  K10 T200 f64" $b --model simple_cart --num-basis 10 --horizon 20.0
run "cfg4  cart  K10 T200 f32" $b --model simple_cart --num-basis 10 --horizon 20.0 --precision f32
done
