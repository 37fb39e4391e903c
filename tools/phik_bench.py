#!/usr/bin/env python3
"""phi_k kernel (Basis::spatialCoeff on a regular grid) against the HBM roofline: times
eea_set_target_grid on large device-resident grids.  Algorithmic bytes = nx*ny*sizeof(real)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ergodic_exploration_amd import capi  # noqa: E402


def occupancy_case(n, K, prec):
    f32 = prec == "occ32"
    eng = capi.Engine(capi.make_config(capi.MODEL_OMNI, 0.1, 2.0, 0.1, 1.0, K, np.eye(3), [-1] * 3, [1] * 3,
                                       precision=capi.PREC_F32 if f32 else capi.PREC_F64))
    g = torch.Generator(device="cuda").manual_seed(2024)
    r = torch.rand((n * n,), device="cuda", generator=g)
    occ = torch.where(r < 0.7, 0, torch.where(r < 0.8, 100, -1)).to(torch.int8)
    del r
    lx = ly = (n - 1) * 0.1
    for _ in range(2):
        eng.set_target_occupancy(n, n, occ, lx, ly)
    torch.cuda.synchronize()
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        eng.set_target_occupancy(n, n, occ, lx, ly)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    pk = eng.phik()
    print("grid %5d^2 K=%2d %s: %8.3f ms  %7.1f GB/s of int8 cells  phik[0]=%.12f phik[1]=%.3e"
          % (n, K, prec, dt * 1e3, n * n / dt / 1e9, pk[0], pk[1]))
    eng.close()
    del occ
    torch.cuda.empty_cache()


def main():
    cases = [(1024, 30, "f64"), (4096, 10, "f64"), (8192, 10, "f64"), (8192, 20, "f64"), (8192, 30, "f64"),
             (8192, 10, "f32"), (16384, 10, "f32"),
             # occupancy input (int8 cells, entropy fused): algorithmic bytes = nx*ny
             (1024, 30, "occ64"), (8192, 10, "occ64"), (16384, 10, "occ64"), (32768, 10, "occ64"),
             (32768, 10, "occ32"), (16384, 30, "occ64")]
    if os.environ.get("PHIK_CASES"):  # e.g. PHIK_CASES="8192:16:f64,8192:17:f64"
        cases = [(int(a), int(b), c) for a, b, c in (x.split(":") for x in os.environ["PHIK_CASES"].split(","))]
    for n, K, prec in cases:
        if prec.startswith("occ"):
            occupancy_case(n, K, prec)
            continue
        f32 = prec == "f32"
        eng = capi.Engine(capi.make_config(capi.MODEL_OMNI, 0.1, 2.0, 0.1, 1.0, K, np.eye(3), [-1] * 3, [1] * 3,
                                           precision=capi.PREC_F32 if f32 else capi.PREC_F64))
        phi = torch.rand((n * n,), dtype=torch.float32 if f32 else torch.float64, device="cuda")
        phi /= phi.sum()
        lx = ly = (n - 1) * 0.1
        for _ in range(2):
            eng.set_target_grid(n, n, phi, lx, ly)
        torch.cuda.synchronize()
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps):
            eng.set_target_grid(n, n, phi, lx, ly)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        nbytes = n * n * (4 if f32 else 8)
        pk = eng.phik()
        print("grid %5d^2 K=%2d %s: %8.3f ms  %7.1f GB/s (%.1f%% of 8 TB/s)  phik[0]=%.12f"
              % (n, K, prec, dt * 1e3, nbytes / dt / 1e9, 100 * nbytes / dt / 8e12, pk[0]))
        eng.close()
        del phi
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
