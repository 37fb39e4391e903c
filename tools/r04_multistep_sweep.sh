#!/bin/bash
# Round 4: receding-horizon steps per launch (eea_control_batch_steps) -- us per pass of 4096 agents by steps per launch
# and agent groups, same box, interleaved with the r03-equivalent library as the anchor.
LEGS="--steps 6 --warmup 2 --passes-per-step 400 --cpu-seconds 0 --no-latency --no-exchange --no-phik --no-grid-tile"
run() { python3 bench.py $LEGS "$@" 2>/dev/null | tail -1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.2f us/pass (groups %d, steps/launch %d, agents %d)" % (1e3*d["ms_per_pass"], d["config"]["agent_groups"], d["config"].get("steps_per_launch",1), d["config"]["agents_per_gpu"]))'; }
for rep in 1 2; do

  for g in 2 1; do for n in 1 2 10 50 400; do echo "[main]           $(run --agent-groups $g --steps-per-launch $n)"; done; done
done
echo "[main] 4 groups:  $(run --agent-groups 4 --steps-per-launch 50)"
echo "[main] omni:      $(run --model omni --steps-per-launch 1)  |  $(run --model omni --steps-per-launch 50)"
