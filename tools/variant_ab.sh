#!/bin/bash
# same-box comparison of library variants (ergodic_exploration_amd/lib/libergodic_amd<variant>.so): headline pass and
# the exchange ladder, interleaved.  usage: tools/variant_ab.sh <outdir> "" _v120s _v120p
out=$1; shift
mkdir -p $out
for rep in 1 2; do
  for v in "$@"; do
    EEA_LIB_VARIANT=$v python3 bench.py --steps 6 --warmup 2 --no-grid-tile --no-exchange --no-latency --no-phik --cpu-seconds 0 \
      > $out/bench_${v:-base}_$rep.json 2> $out/bench_${v:-base}_$rep.err
  done
done
for v in "$@"; do
  EEA_LIB_VARIANT=$v python3 tools/ck_cost.py --quick > $out/ck_${v:-base}.txt 2>&1
done
