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


def main():
    cases = [(1024, 30, "f64"), (4096, 10, "f64"), (8192, 10, "f64"), (8192, 20, "f64"), (8192, 30, "f64"),
             (8192, 10, "f32"), (16384, 10, "f32")]
    for n, K, prec in cases:
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
