#!/bin/bash
# Headline pass time by build variant of control_wave_kernel.o (compiler scheduling flags), same box, two rounds.
for round in 1 2; do
for v in "" _f2 _f3 _f4 _f5 _f6 _f8 _f9 _f10; do
  out=$(EEA_LIB_VARIANT=$v python3 bench.py --steps 5 --warmup 2 --passes-per-step 4000 --cpu-seconds 0 --no-latency --no-exchange --no-phik --no-grid-tile --no-other-configs 2>/dev/null | tail -1)
  echo "round $round variant [${v:-base}]  $(echo "$out" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%6.3f us per pass   %.4f" % (1e3*d["ms_per_pass"], d["roofline"]["frac"]))')"
done
done
