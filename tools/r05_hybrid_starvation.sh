#!/bin/bash
# Hybrid exchange (one agent group device-bound, one stream-ordered) with a collective kernel in the exchange, at full occupancy:
# does the stall follow the collective kernel's FOOTPRINT?  (big = 512 threads x 96 registers x 16 KB LDS; small = one wavefront)
# usage: r05_hybrid_starvation.sh [runs = 10]
F=$PWD/tests/fake_rccl/librccl.so.1
B=ergodic_exploration_amd/host/build/consensus_bench
RUNS=${1:-10}
for agents in 4096 3840; do
  for small in 0 1; do
    stalled=0
    for i in $(seq $RUNS); do
      l=$(FAKE_RCCL_SMALL=$small timeout 200 $B 3000 $agents 1 $F 2 2 2>&1 | grep "consensus every")
      t=$(echo "$l" | sed -n 's/.*agents timed out: \([0-9]*\).*/\1/p')
      us=$(echo "$l" | sed -n 's/.*(bound) *\([0-9.]*\) us per pass.*/\1/p')
      echo "   agents $agents, $([ $small = 1 ] && echo small || echo big) collective kernel, run $i: $us us per pass, agents timed out $t"
      [ "${t:-0}" != "0" ] && stalled=$((stalled + 1))
    done
    echo "== agents $agents, collective kernel footprint $([ $small = 1 ] && echo small || echo big): $stalled of $RUNS runs stalled"
  done
done
