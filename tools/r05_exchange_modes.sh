F=$PWD/tests/fake_rccl/librccl.so.1
make -s -C tests/fake_rccl librccl.so.1
for cfg in "3000 4096 1 1 12" "3000 4096 1 2 12" "3000 4096 1 2 12" "3000 4096 1 3 12" "3000 4096 1 2 2" "3000 4096 1 2 2" "3000 4096 1 3 2" "3000 3968 1 2 2" "3000 3840 1 1 2" "3000 3840 1 2 2"; do
  set -- $cfg
  timeout 120 ergodic_exploration_amd/host/build/consensus_bench $1 $2 $3 $F $4 $5 2>&1 | grep "consensus every" | cut -c1-230 | sed "s/^/[$cfg] /"
done
echo "--- no collective (local), stream-ordered lag 1 / 2"
timeout 120 ergodic_exploration_amd/host/build/consensus_bench 3000 4096 1 "" 1 12 2>&1 | grep "consensus every" | cut -c1-200
timeout 120 ergodic_exploration_amd/host/build/consensus_bench 3000 4096 1 "" 2 12 2>&1 | grep "consensus every" | cut -c1-200
