#!/usr/bin/env python3
"""Cost of the optional c_k output / shared c_k input / exchange waits of eea_control_batch at the metric point
(4096 agents, one launch per pass): none of them moves the pass time (profiles/r02_ablation.txt)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from ergodic_exploration_amd import capi
B=4096
model = capi.MODEL_SIMPLE_CART
lim = np.array([1.0, 0.0, 2.0])
eng = capi.Engine(capi.make_config(model, 0.1, 20.0, 0.1, 1.0, 10, np.diag([1.0, 0.0, 2.0]), -lim, lim))
eng.set_target_gaussians([[2.5, 2.5], [8.5, 2.5]], [[1.5, 1.5], [1.5, 1.5]])
eng.config_domain((-1.0, 11.0, -1.0, 5.0))
T, K2 = eng.T, eng.K2
rng = np.random.default_rng(1)
poses = np.stack([rng.uniform(-0.5, 10.5, B), rng.uniform(-0.5, 4.5, B), rng.uniform(-3, 3, B)], 1)
d_pose = torch.as_tensor(poses).cuda()
d_ut = torch.zeros((B, T, 3), dtype=torch.float64, device="cuda")
d_u0 = torch.empty((B, 3), dtype=torch.float64, device="cuda")
d_ck = torch.empty((B, K2), dtype=torch.float64, device="cuda")
d_cbar = torch.zeros((K2,), dtype=torch.float64, device="cuda")
s = torch.cuda.Stream()
comm = capi.Comm(0, 1, 0, None)
def run(name, f, n=2000):
    for _ in range(200): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(n): f()
    e1.record(s)
    torch.cuda.synchronize()
    print("%-40s %.2f us/pass" % (name, 1e3 * e0.elapsed_time(e1) / n))
run("plain", lambda: eng.control_batch(B, d_pose, d_ut, d_u0, stream=s.cuda_stream))
run("ck out", lambda: eng.control_batch(B, d_pose, d_ut, d_u0, ck=d_ck, stream=s.cuda_stream))
run("ck shared in", lambda: eng.control_batch(B, d_pose, d_ut, d_u0, ck_shared=d_cbar, stream=s.cuda_stream))
run("ck out + shared in", lambda: eng.control_batch(B, d_pose, d_ut, d_u0, ck=d_ck, ck_shared=d_cbar, stream=s.cuda_stream))
comm.consensus_ck_async(eng, B, d_ck, d_cbar, s.cuda_stream, 0)
def f():
    comm.wait(0, s.cuda_stream)
    eng.control_batch(B, d_pose, d_ut, d_u0, ck=d_ck, ck_shared=d_cbar, stream=s.cuda_stream)
run("wait + ck out + shared in", f)
