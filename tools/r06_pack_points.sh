#!/bin/bash
# Quick same-box table of the short-horizon legs (stand-alone runs of bench.py's other_configs legs, tools/other_config_point.py):
# config, agents, lanes per agent, us per pass, us per 4096 agents, fraction of the fp64 vector peak.
P='import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("%-52s %6d agents  L=%2d  %8.3f us/pass  %7.3f us/4096  frac %.4f" % (r["config"], r["agents"], r["lanes_per_agent"], 1e3*r["ms_per_pass"], r["us_per_4096_agents"], r["roofline"]["frac"]))'
for c in "configs[0], chip-filling batch" "configs[1], chip-filling batch" "explore_omni.yaml as shipped, chip-filling batch" "configs[0]" "configs[1]" "explore_omni.yaml as shipped (K = 10, T = 50)"; do
  python3 tools/other_config_point.py --case "$c" 2>/dev/null | python3 -c "$P"
done
for a in 12288 24576 32768; do
  python3 tools/other_config_point.py --case "explore_omni.yaml as shipped (K = 10, T = 50)" --agents $a 2>/dev/null | python3 -c "$P"
done
for a in 16384 24576; do
  python3 tools/other_config_point.py --case "configs[1]" --agents $a --lanes 16 2>/dev/null | python3 -c "$P"
done
