"""Helpers shared by the -m gpu parity tests: synthetic inputs and an
oracle-vs-engine driver.  The oracle is the checker only; the engine is the C ABI."""
import numpy as np

from oracle import pyoracle as po
from ergodic_exploration_amd import capi

MAP_BOUNDS = (-1.0, 11.0, -1.0, 5.0)   # 12 x 6 m domain
MEANS = [[2.5, 2.5], [8.5, 2.5]]       # config/explore_cart.yaml:69-70
SIGMAS = [[1.5, 1.5], [1.5, 1.5]]

MODELS = {"omni": (po.MODEL_OMNI, capi.MODEL_OMNI, [1.0, 1.0, 2.0], [1.0, 1.0, 2.0]),
          "simple_cart": (po.MODEL_SIMPLE_CART, capi.MODEL_SIMPLE_CART, [1.0, 0.0, 2.0], [1.0, 0.0, 2.0])}


def random_poses(rng, B, bounds=MAP_BOUNDS):
    """x ~ U(0.5, lx-0.5) + xmin, y likewise, theta ~ U(-pi, pi)"""
    lx, ly = bounds[1] - bounds[0], bounds[3] - bounds[2]
    p = np.empty((B, 3))
    p[:, 0] = rng.uniform(0.5, lx - 0.5, B) + bounds[0]
    p[:, 1] = rng.uniform(0.5, ly - 0.5, B) + bounds[2]
    p[:, 2] = rng.uniform(-np.pi, np.pi, B)
    return p


def make_pair(model, K, horizon, dt=0.1, resolution=0.1, expl_weight=1.0, precision=capi.PREC_F64,
              means=MEANS, sigmas=SIGMAS, bounds=MAP_BOUNDS, n_oracles=1):
    """(engine, [oracle controllers]) with identical parameters, target and domain."""
    om, em, rdiag, lim = MODELS[model]
    Rinv = np.diag(rdiag)
    lim = np.array(lim)
    eng = capi.Engine(capi.make_config(em, dt, horizon, resolution, expl_weight, K, Rinv, -lim, lim,
                                       precision=precision))
    eng.set_target_gaussians(means, sigmas)
    eng.config_domain(bounds)
    ors = []
    for _ in range(n_oracles):
        o = po.ErgodicControl(om, dt, horizon, resolution, expl_weight, K, Rinv, -lim, lim)
        o.set_target(means, sigmas)
        o.config_target(bounds)
        ors.append(o)
    return eng, ors


def angle_diff(a, b):
    """difference of headings modulo 2 pi (the rollout wraps per step in the reference and
    after the prefix sum in the kernel)"""
    d = np.asarray(a) - np.asarray(b)
    return (d + np.pi) % (2 * np.pi) - np.pi
