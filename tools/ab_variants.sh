#!/bin/bash
# Same-box A/B of library builds (interleaved rounds, one process per measurement).  A variant is a library built
# with `make -C ergodic_exploration_amd/csrc EXTRA=-D... VARIANT=_x` (-> lib/libergodic_amd_x.so); "main" = the
# shipping library.  Run through gpurun.
#   tools/ab_variants.sh "main _x _y" [bench.py arguments, default: the headline shape]
#   ROUNDS=5 tools/ab_variants.sh "main _old" --model omni --num-basis 20 --horizon 5.0 --dt 0.02
VARIANTS=${1:-main}; shift
ROUNDS=${ROUNDS:-3}
for r in $(seq 1 "$ROUNDS"); do
  for v in $VARIANTS; do
    lv=$v; [ "$v" = main ] && lv=""
    out=$(EEA_LIB_VARIANT=$lv python3 bench.py --steps 10 --warmup 3 --passes-per-step 200 --cpu-seconds 0 --no-latency \
          --no-exchange --no-phik --no-grid-tile "$@" 2>/dev/null | tail -1)
    echo "round $r [$v] $(echo "$out" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.4g opt/s  %.2f us/pass  (groups %d)" % (d["value"], 1e3*d["ms_per_pass"], d["config"]["agent_groups"]))')"
  done
done
