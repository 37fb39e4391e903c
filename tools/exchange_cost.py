#!/usr/bin/env python3
"""What the device-bound consensus exchange costs, one piece at a time (one GPU; K = 10, T = 200, 4096 agents, two agent
groups, ONE launch per pass and group): us per pass for
  plain                      eea_control_batch only
  + records                  per-agent sum records out (eea_batch_io::d_ck_rec)
  + ready marks              ... written through, with the agents' ready marks (d_rec_ready)
  + sum record in            ... and a (stale, always ready) sum record consumed (d_ck_shared, ck_shared_parts = 1)
  + bound sum, unconsumed    ... and eea_comm_records_exchange_bound every pass, nobody waiting for its flag
  consensus lag 4 / 2 / 1    the full protocol: pass i waits in-kernel for the flag of pass i - lag
Run on the GPU box: python3 tools/exchange_cost.py [--rccl]   (--rccl: a real one-rank RCCL communicator: sum -> all-reduce
-> publish)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ergodic_exploration_amd import capi  # noqa: E402

B, G, NB, PASSES = 4096, 2, 8, 3000
rccl = "--rccl" in sys.argv
lim = np.array([1.0, 0.0, 2.0])
eng = capi.Engine(capi.make_config(capi.MODEL_SIMPLE_CART, 0.1, 20.0, 0.1, 1.0, 10, np.diag([1.0, 0.0, 2.0]), -lim, lim))
eng.set_target_gaussians([[2.5, 2.5], [8.5, 2.5]], [[1.5, 1.5], [1.5, 1.5]])
eng.config_domain((-1.0, 11.0, -1.0, 5.0))
T, L = eng.T, eng.ck_record_len
rng = np.random.default_rng(12345)
poses = np.stack([rng.uniform(0.5, 11.5, B) - 1.0, rng.uniform(0.5, 5.5, B) - 1.0, rng.uniform(-np.pi, np.pi, B)], 1)
d_pose = torch.as_tensor(poses).cuda()
d_ut = torch.zeros((B, T, 3), dtype=torch.float64, device="cuda")
d_u0 = torch.empty((B, 3), dtype=torch.float64, device="cuda")
arec = [torch.zeros((B, L), dtype=torch.float64, device="cuda") for _ in range(NB)]
rec = [torch.zeros((L,), dtype=torch.float64, device="cuda") for _ in range(NB)]
for r in rec:
    r[100] = 1.0   # a record with one agent: finite consensus for the "stale record" row
ready = torch.zeros((B,), dtype=torch.int32, device="cuda")
flag = torch.zeros((1,), dtype=torch.int32, device="cuda")
status = torch.zeros((B,), dtype=torch.int32, device="cuda")
streams = [torch.cuda.Stream() for _ in range(G)]
gb = [0, B // 2, B]
comm = capi.Comm(0, 1, 0, capi.comm_unique_id() if rccl else None)
seq0 = [0]


def run(label, records=False, marks=False, rec_in=False, exchange=False, lag=None):
    calls = {}
    d_ut.zero_()
    torch.cuda.synchronize()

    def one(i):
        seq, slot = seq0[0] + i + 1, i % NB
        src = (i - lag) % NB if (lag and i >= lag) else None
        for g in range(G):
            key = (g, slot, src)
            c = calls.get(key)
            if c is None:
                sl = slice(gb[g], gb[g + 1])
                kw = {}
                if records:
                    kw["ck_rec"] = arec[slot][sl]
                if marks:
                    kw["rec_ready"], kw["status"] = ready[sl], status[sl]
                if rec_in and not lag:
                    kw["ck_shared"], kw["ck_shared_parts"] = rec[(slot + 3) % NB], 1
                if src is not None:
                    kw["ck_shared"], kw["ck_shared_parts"], kw["ck_flag"] = rec[src], 1, flag
                c = calls[key] = eng.prepared_batch(gb[g + 1] - gb[g], d_pose[sl], d_ut[sl], d_u0[sl],
                                                    stream=streams[g].cuda_stream, **kw)
            if marks or src is not None:
                c(seq, seq - (lag or 0))
            else:
                c()
        if exchange:
            comm.records_exchange_bound(eng, B, arec[slot], ready, seq, rec[slot], flag, slot)

    for i in range(300):
        one(i)
    torch.cuda.synchronize()
    seq0[0] += 400
    t0 = time.perf_counter()
    for i in range(PASSES):
        one(i)
    torch.cuda.synchronize()
    us = 1e6 * (time.perf_counter() - t0) / PASSES
    seq0[0] += PASSES + 8
    print("%-46s %7.2f us per pass   (agents timed out: %d)" % (label, us, int((status != 0).sum().item())))
    return us


print("# device-bound exchange, %s communicator; %d passes per row, wall clock around the loop" % ("one-rank RCCL" if rccl else "local", PASSES))
for rep in range(2):
    base = run("plain (one launch per pass and group)")
    run("+ records out", records=True)
    run("+ ready marks (write-through records)", records=True, marks=True)
    run("+ a stale sum record in", records=True, marks=True, rec_in=True)
    run("+ bound record sum every pass, unconsumed", records=True, marks=True, rec_in=True, exchange=True)
    for lag in (4, 2, 1):
        us = run("consensus, lag %d" % lag, records=True, marks=True, exchange=True, lag=lag)
        print("%46s = %.3f x plain" % ("", us / base))
comm.close()
eng.close()
