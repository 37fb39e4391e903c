LEGS="--steps 6 --warmup 2 --passes-per-step 400 --cpu-seconds 0 --no-latency --no-exchange --no-phik --no-grid-tile"
run() { python3 bench.py $LEGS "$@" 2>/dev/null | tail -1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.2f us/pass (groups %d, steps/launch %d)" % (1e3*d["ms_per_pass"], d["config"]["agent_groups"], d["config"].get("steps_per_launch",1)))'; }
for rep in 1 2; do for v in "" _stag13 _stag6; do for g in 1 2; do for n in 20 50 100; do echo "[${v:-main}] $(EEA_LIB_VARIANT=$v run --agent-groups $g --steps-per-launch $n)"; done; done; done; done
