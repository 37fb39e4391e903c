#!/usr/bin/env python3
"""Throughput of the kernels that follow control() in every tick of the Exploration loop
(SURVEY.md 8(f) ranks 1-2): batched collision check, validate_control and DynamicWindow::control
(both overloads) on a 1200 x 600 cell map (120 x 60 m at 0.1 m) with obstacle blocks."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ergodic_exploration_amd import capi  # noqa: E402

COLL = (0.7, 1.0, 0.2, 0.8)
DWA_OMNI = (0.1, 1.0, 0.2, 1.0, 1.0, 1.0, 1.0, -1.0, 1.0, -1.0, 2.0, -2.0, 3, 8, 5)
DWA_CART = (0.1, 2.0, 0.2, 2.5, 0.0, 1.0, 1.0, -1.0, 0.0, 0.0, 2.0, -2.0, 3, 1, 5)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    rng = np.random.default_rng(1)
    xs, ys, res = 1200, 600, 0.1
    data = np.zeros((ys, xs), dtype=np.int8)
    for _ in range(300):
        i, j = rng.integers(0, ys - 12), rng.integers(0, xs - 12)
        data[i:i + rng.integers(2, 12), j:j + rng.integers(2, 12)] = 100
    data[rng.integers(0, ys, 4000), rng.integers(0, xs, 4000)] = -1
    ccfg = capi.make_collision_cfg(0.0, 0.0, res, xs, ys, *COLL)
    d_grid = torch.as_tensor(data).cuda()
    sizes = [int(v) for v in os.environ.get("TICK_SIZES", "4096,65536").split(",")]
    for P in sizes:
        x0 = np.stack([rng.uniform(2, 118, P), rng.uniform(2, 58, P), rng.uniform(-np.pi, np.pi, P)], 1)
        vb = np.stack([rng.uniform(-1, 1, P), rng.uniform(-1, 1, P), rng.uniform(-2, 2, P)], 1)
        d_x0, d_vb = torch.as_tensor(x0).cuda(), torch.as_tensor(vb).cuda()
        d_hit = torch.empty((P,), dtype=torch.int32, device="cuda")
        d_u = torch.empty((P, 3), dtype=torch.float64, device="cuda")
        d_f = torch.empty((P,), dtype=torch.int32, device="cuda")
        t = timed(lambda: capi.collision_check_batch(ccfg, d_grid, d_x0, d_hit))
        print("P=%6d collision_check      %8.1f us  %.3g poses/s  (hits %.1f %%)" % (P, t * 1e6, P / t, 100 * d_hit.float().mean().item()))
        t = timed(lambda: capi.validate_control_batch(ccfg, d_grid, d_x0, d_vb, 0.1, 0.5, d_hit))
        print("P=%6d validate_control     %8.1f us  %.3g twists/s (valid %.1f %%)" % (P, t * 1e6, P / t, 100 * d_hit.float().mean().item()))
        for name, dwa in (("omni 3x8x5", DWA_OMNI), ("cart 3x1x5", DWA_CART)):
            cfg = capi.DwaCfg(*dwa)
            t = timed(lambda: capi.dwa_control_batch(ccfg, cfg, d_grid, d_x0, d_vb, d_u, d_f, vref=d_vb), reps=5)
            print("P=%6d dwa vref %-11s %8.1f us  %.3g plans/s  (found %.1f %%)" % (P, name, t * 1e6, P / t, 100 * d_f.float().mean().item()))
        n_ref = 50
        xt = np.repeat(x0[:, None, :], n_ref, 1) + rng.normal(0, 0.2, (P, n_ref, 3))
        d_xt = torch.as_tensor(xt).cuda()
        cfg = capi.DwaCfg(*DWA_OMNI)
        t = timed(lambda: capi.dwa_control_batch(ccfg, cfg, d_grid, d_x0, d_vb, d_u, d_f, xt_ref=d_xt, dt_ref=0.1), reps=5)
        print("P=%6d dwa traj omni 3x8x5  %8.1f us  %.3g plans/s  (found %.1f %%)" % (P, t * 1e6, P / t, 100 * d_f.float().mean().item()))


if __name__ == "__main__":
    main()
