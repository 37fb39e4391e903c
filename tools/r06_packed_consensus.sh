#!/bin/bash
# Round 6: the consensus pass on the several-agents-per-wavefront kernel (explore_omni.yaml's T = 50; configs[1]'s T = 20), per-agent
# sum records against one record per wavefront (eea_batch_io::rec_per_wavefront).  Output: stdout.
cd "$(dirname "$0")/.."
B=ergodic_exploration_amd/host/build/consensus_bench
F=$PWD/tests/fake_rccl/librccl.so.1
run() { # label, horizon, model, agents, groups/mode (2 = device-bound, 32 = gated), lag, wave records, collective library
  echo "-- $1"
  CONSENSUS_BENCH_HORIZON=$2 CONSENSUS_BENCH_MODEL=$3 CONSENSUS_BENCH_WAVE_RECORDS=$7 timeout 300 $B 3000 $4 1 "$8" $6 $5 2>&1 | grep "consensus every\|plain passes  \|records through" | cut -c1-230
}
for agents in 12288 32768; do
  echo "== explore_omni.yaml (T = 50, omni), $agents agents"
  for wave in 0 1; do
    run "device-bound lag 1, wave records $wave" 5.0 omni $agents 2 1 $wave ""
    run "gated lag 2 (local), wave records $wave" 5.0 omni $agents 32 2 $wave ""
    run "gated lag 2 + collective kernel, wave records $wave" 5.0 omni $agents 32 2 $wave $F
  done
done
echo "== configs[1] (T = 20, SimpleCart), 32768 agents"
for wave in 0 1; do
  run "device-bound lag 1, wave records $wave" 2.0 cart 32768 2 1 $wave ""
  run "gated lag 2 + collective kernel, wave records $wave" 2.0 cart 32768 32 2 $wave $F
done
echo "== soak: explore_omni.yaml, 32768 agents, gated lag 2 + collective kernel, one record per wavefront, 60 000 passes"
CONSENSUS_BENCH_HORIZON=5.0 CONSENSUS_BENCH_MODEL=omni CONSENSUS_BENCH_WAVE_RECORDS=1 timeout 600 $B 60000 32768 1 "$F" 2 32 2>&1 | grep "consensus every" | cut -c1-230
