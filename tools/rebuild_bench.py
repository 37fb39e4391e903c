#!/usr/bin/env python3
"""Wall time of a phi_k rebuild through eea_config_domain (configTarget with a changed map extent:
Target::fill + normalisation + Basis::spatialCoeff on the device), Gaussian target."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ergodic_exploration_amd import capi  # noqa: E402

for K, lx, ly in ((10, 12.0, 6.0), (20, 25.5, 25.5), (30, 102.3, 102.3)):
    eng = capi.Engine(capi.make_config(capi.MODEL_OMNI, 0.1, 2.0, 0.1, 1.0, K, np.eye(3), [-1] * 3, [1] * 3))
    eng.set_target_gaussians([[2.5, 2.5], [8.5, 2.5]], [[1.5, 1.5], [1.5, 1.5]])
    for i in range(3):
        eng.config_domain((0.0, lx + 0.1 * i, 0.0, ly))
    n = 50
    t0 = time.perf_counter()
    for i in range(n):
        eng.config_domain((0.0, lx + 0.1 * (i % 2), 0.0, ly))  # the extent changes every call
    dt = (time.perf_counter() - t0) / n
    print("K=%2d grid %4d x %4d: %7.1f us per rebuild" % (K, round(lx / 0.1) + 1, round(ly / 0.1) + 1, dt * 1e6))
    eng.close()
