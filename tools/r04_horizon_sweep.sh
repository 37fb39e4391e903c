#!/bin/bash
# Pass time and roofline fraction over the horizon at K = 10 (fp64, SimpleCart, 4096 agents, headline launch form): where the
# lane map changes shape (64 lanes x 1..4 steps; T = 193 .. 200 top-heavy with the cooperative last slot).
for T in 20 50 64 100 128 150 192 196 200 208 224 250 256; do
  h=$(python3 -c "print($T * 0.1)")
  out=$(python3 bench.py --steps 5 --warmup 2 --passes-per-step 1000 --cpu-seconds 0 --no-latency --no-exchange --no-phik --no-grid-tile --no-other-configs --horizon $h 2>/dev/null | tail -1)
  echo "T = $T  $(echo "$out" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%6.2f us per pass   %.3f of the fp64 vector peak   %.2f us per 64 steps" % (1e3*d["ms_per_pass"], d["roofline"]["frac"], 1e3*d["ms_per_pass"]*64/d["config"]["horizon_steps"]))')"
done
