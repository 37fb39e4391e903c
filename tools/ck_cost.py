#!/usr/bin/env python3
"""Where the consensus pass spends its time at the metric point (4096 agents): the optional record output / sum-record
input of eea_control_batch, the one-launch record sum, the cross-stream choreography, each added in turn.  Two agent
groups on two streams as in bench.py (profiles/r03_ablation.txt)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from ergodic_exploration_amd import capi
B, G = 4096, 2
model = capi.MODEL_SIMPLE_CART
lim = np.array([1.0, 0.0, 2.0])
eng = capi.Engine(capi.make_config(model, 0.1, 20.0, 0.1, 1.0, 10, np.diag([1.0, 0.0, 2.0]), -lim, lim))
eng.set_target_gaussians([[2.5, 2.5], [8.5, 2.5]], [[1.5, 1.5], [1.5, 1.5]])
eng.config_domain((-1.0, 11.0, -1.0, 5.0))
T, K2, L = eng.T, eng.K2, eng.ck_record_len
rng = np.random.default_rng(1)
poses = np.stack([rng.uniform(-0.5, 10.5, B), rng.uniform(-0.5, 4.5, B), rng.uniform(-3, 3, B)], 1)
d_pose = torch.as_tensor(poses).cuda()
d_ut = torch.zeros((B, T, 3), dtype=torch.float64, device="cuda")
d_u0 = torch.empty((B, 3), dtype=torch.float64, device="cuda")
NB = 8
d_arec = [torch.zeros((B, L), dtype=torch.float64, device="cuda") for _ in range(NB)]
d_rec = [torch.zeros((L,), dtype=torch.float64, device="cuda") for _ in range(NB)]
for r in d_rec:
    r[K2] = 1.0
comm = capi.Comm(0, 1, 0, None)   # local communicator: its own highest-priority exchange stream, no RCCL
streams = [torch.cuda.Stream() for _ in range(G)]
xs = torch.cuda.Stream()   # the torch-stream variants below: normal priority first, highest priority afterwards
gb = [(g * B) // G for g in range(G + 1)]
ev_g = [[torch.cuda.Event() for _ in range(G)] for _ in range(NB)]
ev_x = [torch.cuda.Event() for _ in range(NB)]
calls = {}


def call(g, rec_slot=None, src=None):
    key = (g, rec_slot, src)
    if key not in calls:
        sl = slice(gb[g], gb[g + 1])
        calls[key] = eng.prepared_batch(gb[g + 1] - gb[g], d_pose[sl], d_ut[sl], d_u0[sl], stream=streams[g].cuda_stream,
                                        ck_rec=None if rec_slot is None else d_arec[rec_slot][sl],
                                        ck_shared=None if src is None else d_rec[src],
                                        ck_shared_parts=0 if src is None else 1)
    calls[key]()


def run(name, f, n=3000):
    for i in range(400):
        f(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(streams[0])
    import time
    t0 = time.perf_counter()
    for i in range(n):
        f(i)
    enq = time.perf_counter() - t0
    comm.flush()
    for g in range(1, G):
        e = torch.cuda.Event()
        e.record(streams[g])
        streams[0].wait_event(e)
    e1.record(streams[0])
    torch.cuda.synchronize()
    print("%-58s %.2f us/pass  (host enqueue %.1f us/pass)" % (name, 1e3 * e0.elapsed_time(e1) / n, 1e6 * enq / n))


def plain(i):
    for g in range(G):
        call(g)


def rec_out(i):
    for g in range(G):
        call(g, i % NB)


def rec_in(i):
    for g in range(G):
        call(g, None, 0)


def rec_in_out(i):
    for g in range(G):
        call(g, i % NB, (i - 2) % NB)


def with_sum_unordered(i):   # the sum launch on the exchange stream, nothing waits for anything
    for g in range(G):
        call(g, i % NB, (i - 2) % NB)
    eng.ck_records_sum(B, d_arec[i % NB], d_rec[i % NB], stream=xs.cuda_stream)


def with_tiny_sum_unordered(i):   # the same launch with one agent: what a third queue costs by itself
    for g in range(G):
        call(g, i % NB, (i - 2) % NB)
    eng.ck_records_sum(1, d_arec[i % NB], d_rec[(i + 4) % NB], stream=xs.cuda_stream)


def with_sum_after_groups(i):  # the exchange stream waits for the groups' launches; consumers do not wait
    s = i % NB
    for g in range(G):
        call(g, s, (i - 2) % NB)
        ev_g[s][g].record(streams[g])
        xs.wait_event(ev_g[s][g])
    eng.ck_records_sum(B, d_arec[s], d_rec[s], stream=xs.cuda_stream)
    ev_x[s].record(xs)


def full(i, lag=2):
    s = i % NB
    src = (i - lag) % NB
    for g in range(G):
        if i >= lag:
            streams[g].wait_event(ev_x[src])
        call(g, s, src)
        ev_g[s][g].record(streams[g])
    for g in range(G):
        xs.wait_event(ev_g[s][g])
    eng.ck_records_sum(B, d_arec[s], d_rec[s], stream=xs.cuda_stream)
    ev_x[s].record(xs)


xc = {}


def full_c(i, lag=3, NBc=6):   # the same through ONE C-ABI call per pass (eea_comm_records_exchange_async) + eea_comm_wait
    s, src = i % NBc, (i - lag) % NBc
    for g in range(G):
        if i >= lag:
            w = xc.get(("w", g, src))
            if w is None:
                w = xc[("w", g, src)] = comm.prepared_wait(src, streams[g].cuda_stream)
            w()
        call(g, s, src)
    x = xc.get(s)
    if x is None:
        x = xc[s] = comm.prepared_records_exchange(eng, B, d_arec[s], d_rec[s], [st.cuda_stream for st in streams], s)
    x()


def full_c2(i, lag=3, NBc=8, events=True, exchange=True, wait=True):
    """two C-ABI calls per pass: eea_comm_control_groups + eea_comm_records_exchange_async; the flags take pieces out:
    events = the groups' kernels carry completion events, exchange = the exchange call is made, wait = the groups'
    launches wait for the exchange they consume"""
    s, src = i % NBc, (i - lag) % NBc
    use = i >= lag
    key = ("g", s, src if use else None, events, wait)
    c = xc.get(key)
    if c is None:
        groups = [dict(B=gb[g + 1] - gb[g], pose=d_pose[gb[g]:gb[g + 1]], ut=d_ut[gb[g]:gb[g + 1]], u0=d_u0[gb[g]:gb[g + 1]],
                       stream=streams[g].cuda_stream, ck_rec=d_arec[s][gb[g]:gb[g + 1]],
                       ck_shared=d_rec[src] if use else None, ck_shared_parts=1 if use else 0) for g in range(G)]
        c = xc[key] = comm.prepared_control_groups(eng, groups, src if (use and wait) else -1, s if events else -1)
    c()
    if not exchange:
        return
    x = xc.get(s)
    if x is None:
        x = xc[s] = comm.prepared_records_exchange(eng, B, d_arec[s], d_rec[s], [st.cuda_stream for st in streams], s)
    x()


QUICK = "--quick" in sys.argv
run("plain (two groups)", plain)
if QUICK:
    run("+ both", rec_in_out)
    xs = torch.cuda.Stream(priority=-1)
    run("+ record sum on the exchange stream, unordered [highest stream priority]", with_sum_unordered)
    run("+ a ONE-agent record sum on the exchange stream, unordered [highest stream priority]", with_tiny_sum_unordered)
    for nb in (256, 1024, 2048):
        run("+ record sum of the first %d agents only, unordered [highest]" % nb,
            lambda i, nb=nb: (rec_in_out(i), eng.ck_records_sum(nb, d_arec[i % NB], d_rec[i % NB], stream=xs.cuda_stream)))
    run("control_groups only, kernels carry completion events", lambda i: full_c2(i, 3, exchange=False, wait=False))
    run("+ exchange (ordered after the groups), nobody waits for it", lambda i: full_c2(i, 3, wait=False))
    for lag in (2, 3, 4):
        run("two C calls per pass (control_groups + exchange), lag %d" % lag, lambda i: full_c2(i, lag))
    sys.exit(0)
run("+ per-agent records out", rec_out)
run("+ sum record in (ck_shared_parts = 1)", rec_in)
run("+ both", rec_in_out)
for prio, label in ((0, "normal stream priority"), (-1, "highest stream priority")):
    xs = torch.cuda.Stream(priority=prio)
    run("+ record sum on the exchange stream, unordered [%s]" % label, with_sum_unordered)
    run("+ exchange stream ordered after the groups (torch events) [%s]" % label, with_sum_after_groups)
    run("+ consumers wait for the exchange, lag 2 (torch events) [%s]" % label, full)
    run("  the same with lag 3 [%s]" % label, lambda i: full(i, 3))
print("through the C ABI (the communicator's own highest-priority stream, completion events bound to the kernels):")
run("one C call per pass (records_exchange_async), lag 2", lambda i: full_c(i, 2))
run("one C call per pass (records_exchange_async), lag 3", lambda i: full_c(i, 3))
run("one C call per pass (records_exchange_async), lag 4", lambda i: full_c(i, 4))
run("control_groups only, plain launches (no events, no exchange)", lambda i: full_c2(i, 3, events=False, exchange=False, wait=False))
run("control_groups only, kernels carry completion events", lambda i: full_c2(i, 3, exchange=False, wait=False))
run("+ exchange (ordered after the groups), nobody waits for it", lambda i: full_c2(i, 3, wait=False))
run("two C calls per pass (control_groups + exchange), lag 2", lambda i: full_c2(i, 2))
run("two C calls per pass (control_groups + exchange), lag 3", lambda i: full_c2(i, 3))
run("two C calls per pass (control_groups + exchange), lag 4", lambda i: full_c2(i, 4))
comm.host_thread(True)
xc.clear()
print("the same with the exchange's HIP calls on the communicator's host thread (eea_comm_host_thread):")
run("two C calls per pass + host thread, lag 2", lambda i: full_c2(i, 2))
run("two C calls per pass + host thread, lag 3", lambda i: full_c2(i, 3))
run("two C calls per pass + host thread, lag 4", lambda i: full_c2(i, 4))
run("two C calls per pass + host thread, lag 5", lambda i: full_c2(i, 5))
run("two C calls per pass + host thread, lag 6", lambda i: full_c2(i, 6))
comm.host_thread(False)
