#!/usr/bin/env python3
"""eea_set_target_occupancy on the 1024 x 1024 grid of BASELINE configs[4] (K = 30): wall time per call (the call waits for its
stream: phi_k is installed when it returns), next to the row-tiled form of bench.py's grid_tile leg on one rank (rows -> sums ->
eea_set_phik_from_sums: three launches)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ergodic_exploration_amd import capi
n, K, res = 1024, 30, 0.1
l = (n - 1) * res
rng = np.random.default_rng(2024)
blocks = rng.choice(np.array([0, 100, -1], dtype=np.int8), size=(n // 32 + 1, n // 32 + 1), p=[0.7, 0.1, 0.2])
occ = np.ascontiguousarray(np.kron(blocks, np.ones((32, 32), dtype=np.int8))[:n, :n])
d = torch.as_tensor(occ).cuda()
e = capi.Engine(capi.make_config(capi.MODEL_OMNI, 0.1, 50.0, res, 1.0, K, np.diag([1.0, 1.0, 2.0]), [-1.0, -1.0, -2.0], [1.0, 1.0, 2.0]))
for _ in range(20):
    e.set_target_occupancy(n, n, d, l, l)
t0 = time.perf_counter()
for _ in range(300):
    e.set_target_occupancy(n, n, d, l, l)
a = (time.perf_counter() - t0) / 300
p1 = e.phik().copy()
sums = torch.empty((K * K,), dtype=torch.float64, device="cuda")
for _ in range(20):
    e.spatial_coeff_occupancy_rows(n, n, 0, n, d, l, l, sums); e.set_phik_from_sums(sums, l, l); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(300):
    e.spatial_coeff_occupancy_rows(n, n, 0, n, d, l, l, sums); e.set_phik_from_sums(sums, l, l); torch.cuda.synchronize()
b = (time.perf_counter() - t0) / 300
p2 = e.phik().copy()
print("eea_set_target_occupancy (stream + reduction-with-normalisation, then the host wait)  %.2f us per call" % (1e6 * a))
print("rows -> sums -> eea_set_phik_from_sums (three launches) + host wait                   %.2f us per call" % (1e6 * b))
print("phi_k of the two forms bitwise equal:", bool(np.array_equal(p1, p2)), " phi_k[0] =", p1[0])
