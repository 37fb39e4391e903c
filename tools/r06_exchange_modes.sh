#!/bin/bash
# Round 6: what a consensus on EVERY pass costs from a C++ host loop (host/test/consensus_bench.cpp, C ABI),
# 4096 agents, K = 10, T = 200, two agent groups -- per protocol: the pass time, the host's own share, its split over the call
# types, time-outs; with and without a collective KERNEL in the exchange (one rank whose all-reduce is the kernel-shaped test
# double tests/fake_rccl); a 60 000-pass soak of the gated exchange; the packed kernel at yaml T = 50; and the host cost of the
# runtime calls themselves (tools/ubench/host_calls.hip).  Run on the GPU box: tools/r06_exchange_modes.sh > profiles/r06_exchange_modes.txt
F=$PWD/tests/fake_rccl/librccl.so.1
B=ergodic_exploration_amd/host/build/consensus_bench
export CONSENSUS_BENCH_BREAKDOWN=1
run() { timeout 300 $B "$@" 2>&1 | grep "consensus every\|host time\|plain passes  " | cut -c1-250; }
echo "== collective kernel in the exchange (test double: 512 threads x 96 registers x 16 KB LDS per block), lag 2"
echo "-- gated (eea_stream_wait_flag + eea_comm_records_exchange_bound; ABI 6)"; run 6000 4096 1 $F 2 32
echo "-- gated, lag 3"; run 6000 4096 1 $F 3 32
echo "-- one device graph per 48 passes (eea_consensus_plan; ABI 6)"; run 6000 4096 1 $F 2 22
echo "-- one device graph per 192 passes"; CONSENSUS_PLAN_PASSES=192 run 6000 4096 1 $F 2 22
echo "-- stream-ordered per call (eea_comm_records_exchange_async + eea_comm_wait; round 5's rule)"; run 6000 4096 1 $F 2 12
echo "== no collective (local communicator)"
echo "-- device-bound, lag 1 (in-kernel flag wait)"; run 6000 4096 1 "" 1 2
echo "-- gated, lag 2"; run 6000 4096 1 "" 2 32
echo "-- one device graph per 192 passes, lag 2"; CONSENSUS_PLAN_PASSES=192 run 6000 4096 1 "" 2 22
echo "== soak: gated, collective kernel in the exchange, lag 2, 60 000 passes"
run 60000 4096 1 $F 2 32
echo "== host cost of the runtime calls (enqueue only, onto busy streams)"
timeout 60 tools/ubench/host_calls
