# Round 5, VERDICT r04 item 2: the multi-rank exchange paths through the stream-asynchronous, kernel-shaped test double of RCCL.
#  (1) the double by itself: ranks as threads and as processes (tests/fake_rccl/selftest)
#  (2) host/test/rank_tests: two ranks as two processes on the one GPU
#  (3) host/test/consensus_bench: ONE process (the production shape: one process per GPU) with a collective KERNEL in its
#      device-bound exchange -- does it land beside 4096 agents' worth of control wavefronts?  all groups device-bound vs one
#      group stream-ordered (eea_comm_wait), lag 1 / 2 / 3, 4096 and 3968 agents; then two processes on the one GPU
set -u
make -s -C tests/fake_rccl librccl.so.1 selftest
mkdir -p gpurun_out/two_rank
F=$PWD/tests/fake_rccl/librccl.so.1
{
echo "== (1) the double by itself"
for mode in "2 0" "2 2" "2 0 procs" "2 2 procs" "4 2 procs"; do echo "selftest $mode"; timeout 20 tests/fake_rccl/selftest $F $mode; echo "rc=$?"; done
echo "== (2) rank_tests (two processes)"
LD_LIBRARY_PATH=$PWD/tests/fake_rccl:${LD_LIBRARY_PATH:-} timeout 300 ergodic_exploration_amd/host/build/rank_tests | grep -v "^XR "; echo "rc=$?"
echo "== (3) consensus_bench: passes agents ranks lag groups(<0: all groups device-bound)"
for cfg in "2000 4096 1 2 2 60" "2000 4096 1 3 2 60" "2000 4096 1 1 2 60" "2000 3968 1 1 2 60" "2000 4096 1 2 -2 90" "20 4096 1 1 -2 90" "500 2048 2 2 2 90" "500 4096 2 2 2 90"; do
  set -- $cfg
  echo "-- consensus_bench $1 $2 $3 <double> $4 $5"
  timeout $6 ergodic_exploration_amd/host/build/consensus_bench $1 $2 $3 $F $4 $5 2>&1 | grep -v "^RESULT\|hipGraph"; echo "rc=$?"
done
} > gpurun_out/two_rank/r05_two_ranks.txt 2>&1
tail -60 gpurun_out/two_rank/r05_two_ranks.txt
