#!/bin/bash
# launch time vs batch size: separates the fixed part of a launch from the per-agent cost
for a in 256 1024 2048 4096 8192 16384 32768; do
  out=$(python3 bench.py --steps 10 --warmup 3 --passes-per-step 200 --cpu-seconds 0 --no-latency --no-exchange --no-phik --no-grid-tile --agents $a "$@" 2>/dev/null | tail -1)
  echo "agents $a $(echo "$out" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.4g opt/s  launch %.2f us" % (d["value"], 1e3*d["roofline"]["launch_ms"]))')"
done
