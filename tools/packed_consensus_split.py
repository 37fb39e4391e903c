#!/usr/bin/env python3
"""Round 6: where the consensus overhead of a packed short-horizon pass sits.  explore_omni.yaml's shape (T = 50, K = 10, omni),
two agent groups on two streams, passes enqueued back to back; each line adds ONE ingredient of the device-bound consensus pass:
  plain | + records out (per agent / per wavefront) | + ready marks (write-through) | + the record sum running beside |
  + the shared c_k consumed (a static record) | + the flag wait in the kernel (flag already there)
Output: us per pass.   tools/packed_consensus_split.py [agents]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ergodic_exploration_amd import capi

B = int(sys.argv[1]) if len(sys.argv) > 1 else 12288
lim = np.array([1., 1., 2.])
eng = capi.Engine(capi.make_config(capi.MODEL_OMNI, 0.1, 5.0, 0.1, 1.0, 10, np.diag([1., 1., 2.]), -lim, lim))
eng.set_target_gaussians([[2.5, 2.5], [8.5, 2.5]], [[1.5, 1.5], [1.5, 1.5]])
eng.config_domain((-1, 11, -1, 5))
T, RL, K2 = eng.T, eng.ck_record_len, eng.K2
rng = np.random.default_rng(1)
pose = torch.as_tensor(np.stack([rng.uniform(-.5, 10.5, B), rng.uniform(-.5, 4.5, B), rng.uniform(-3, 3, B)], 1)).cuda()
ut = torch.as_tensor(rng.uniform(-.3, .3, (B, T, 3))).cuda()
u0 = torch.empty((B, 3), dtype=torch.float64, device="cuda")
gb = [0, B // 2, B]
st = [torch.cuda.Stream(), torch.cuda.Stream()]
xs = torch.cuda.Stream()
print("agents %d, lanes per agent %d" % (B, eng.agent_lanes(B // 2)))


def run(label, wave=False, rec=False, marks=False, do_sum=False, shared=False, flag=False, passes=400):
    cnt = [eng.record_count(gb[g + 1] - gb[g]) if wave else gb[g + 1] - gb[g] for g in range(2)]
    off = [0, cnt[0], cnt[0] + cnt[1]]
    n_rec = off[2]
    arec = torch.zeros((n_rec, RL), dtype=torch.float64, device="cuda")
    ready = torch.zeros((n_rec,), dtype=torch.int32, device="cuda")
    dsum = torch.zeros((RL,), dtype=torch.float64, device="cuda")
    static = torch.zeros((RL,), dtype=torch.float64, device="cuda")
    static[K2] = 1.0
    fl = torch.full((1,), 1 << 30, dtype=torch.int32, device="cuda")
    def one(seq):
        for g in range(2):
            sl = slice(gb[g], gb[g + 1])
            kw = dict(stream=st[g].cuda_stream)
            if rec:
                kw.update(ck_rec=arec[off[g]:off[g + 1]], rec_per_wavefront=wave)
            if marks:
                kw.update(rec_ready=ready[off[g]:off[g + 1]], rec_seq=seq)
            if shared:
                kw.update(ck_shared=static, ck_shared_parts=1)
            if flag:
                kw.update(ck_flag=fl, ck_flag_seq=1)
            eng.control_batch(gb[g + 1] - gb[g], pose[sl], ut[sl], u0[sl], **kw)
        if do_sum:
            if marks:
                eng.ck_records_sum_bound(n_rec, arec, ready, seq, dsum, None, stream=xs.cuda_stream)
            else:
                eng.ck_records_sum(n_rec, arec, dsum, stream=xs.cuda_stream)
    for i in range(50):
        one(i + 1)
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for i in range(passes):
        one(100 + i)
    torch.cuda.synchronize()
    print("%-78s %7.2f us per pass" % (label, 1e6 * (time.perf_counter() - t0) / passes))


run("plain")
for wave in (False, True):
    w = "one per wavefront" if wave else "one per agent"
    run("+ records out, %s" % w, wave, rec=True)
    run("+ ready marks (records write-through)", wave, rec=True, marks=True)
    run("+ the record sum beside (polls the marks)", wave, rec=True, marks=True, do_sum=True)
    run("+ a shared record consumed (static)", wave, rec=True, marks=True, do_sum=True, shared=True)
    run("+ the flag wait in the kernel (flag already there)", wave, rec=True, marks=True, do_sum=True, shared=True, flag=True)
run("plain + a shared record consumed (static), nothing out", shared=True)
