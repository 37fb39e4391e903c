#!/usr/bin/env python3
"""Writes the per-stage golden fixtures tests/golden/stages_<config>.npz.

The generator is the CPU ORACLE (oracle/ergodic_oracle.c), run here on the configurations SURVEY.md 8(c) lists:
three consecutive control() calls (warm-start shift) per configuration, replay memory 0 / 7 / 100 columns, and
for every call the inputs (pose, controls before the call, memory columns) and the stage outputs traj, ck, edx,
bdx, rhot, ut, u0, plus phi_k of the configuration.  The fixtures are DATA (inputs and expected outputs); two
checkers read them:
  * tests/test_oracle_pinning.py: the independent numpy restatement (tests/np_restatement.py) must reproduce
    every stage to 1e-12 -- a transcription slip in either restatement shows up at the stage where it happens;
  * tests/test_gpu_golden_stages.py: the HIP path must reproduce them on the GPU box.
The reference itself cannot be built in this image (Armadillo and ROS headers absent), so
these are oracle outputs, not reference outputs; the reference-held vectors are tests/golden/reference_kats.json
and survey_anchors.json.

  python tools/gen_golden.py          # rewrites tests/golden/stages_*.npz (deterministic)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
YAML_MEANS, YAML_SIGMAS = [[2.5, 2.5], [8.5, 2.5]], [[1.5, 1.5], [1.5, 1.5]]
MAP12x6 = (-1.0, 11.0, -1.0, 5.0)

# name: model, K, dt, horizon, Rinv diag, limits, bounds, means, sigmas, n_mem   
CONFIGS = {
    "cfg1_omni_K5_T5": ("omni", 5, 0.1, 0.5, [1, 1, 2], [1, 1, 2], MAP12x6, [[2.5, 2.5]], [[1.5, 1.5]], 0),
    "yaml_omni_K10_T50": ("omni", 10, 0.1, 5.0, [1, 1, 2], [1, 1, 2], MAP12x6, YAML_MEANS, YAML_SIGMAS, 7),
    "yaml_cart_K10_T50": ("simple_cart", 10, 0.1, 5.0, [1, 0, 2], [1, 0, 2], MAP12x6, YAML_MEANS, YAML_SIGMAS, 100),
    "cfg2_cart_K10_T20": ("simple_cart", 10, 0.1, 2.0, [1, 0, 2], [1, 0, 2], MAP12x6, YAML_MEANS, YAML_SIGMAS, 0),
    "metric_cart_K10_T200": ("simple_cart", 10, 0.1, 20.0, [1, 0, 2], [1, 0, 2], MAP12x6, YAML_MEANS, YAML_SIGMAS, 0),
    "metric_omni_K10_T200_mem": ("omni", 10, 0.1, 20.0, [1, 1, 2], [1, 1, 2], MAP12x6, YAML_MEANS, YAML_SIGMAS, 100),
    "cfg3_omni_K20_T250": ("omni", 20, 0.02, 5.0, [1, 1, 2], [1, 1, 2], (0.0, 25.5, 0.0, 25.5),
                           [[6.0, 6.0], [19.0, 12.0]], [[3.0, 3.0], [3.0, 3.0]], 0),
}
MODELS = {"omni": po.MODEL_OMNI, "simple_cart": po.MODEL_SIMPLE_CART}


def generate(name):
    model, K, dt, horizon, rdiag, lim, bounds, means, sigmas, n_mem = CONFIGS[name]
    rng = np.random.default_rng(sum(map(ord, name)))  # deterministic per configuration
    lim = np.array(lim, dtype=float)
    ctl = po.ErgodicControl(MODELS[model], dt, horizon, 0.1, 1.0, K, np.diag(np.array(rdiag, dtype=float)), -lim, lim)
    ctl.set_target(means, sigmas)
    ctl.config_target(bounds)
    T = ctl.T
    lx, ly = bounds[1] - bounds[0], bounds[3] - bounds[2]
    x = np.array([bounds[0] + rng.uniform(0.5, lx - 0.5), bounds[2] + rng.uniform(0.5, ly - 0.5), rng.uniform(-3, 3)])
    ut0 = rng.uniform(-0.5, 0.5, (3, T))
    if model == "simple_cart":
        ut0[1] = 0.0
    mem = None
    if n_mem:
        mem = np.vstack([bounds[0] + rng.uniform(0.5, lx - 0.5, n_mem), bounds[2] + rng.uniform(0.5, ly - 0.5, n_mem),
                         rng.uniform(-3, 3, n_mem)])
    ctl.ut = ut0
    out = {"phik": ctl.phik, "lamdak": ctl.lamdak, "bounds": np.array(bounds), "pose": x,
           "mem_cols": mem if mem is not None else np.zeros((3, 0)),
           "params": np.array([dt, horizon, 0.1, 1.0, K], dtype=float), "Rinv_diag": np.array(rdiag, dtype=float),
           "limits": lim, "means": np.array(means, dtype=float), "sigmas": np.array(sigmas, dtype=float)}
    for call in range(3):
        out["ut_in_%d" % call] = ctl.ut
        u, st = ctl.control(bounds, x, mem, stages=True)
        for k, v in st.items():
            out["%s_%d" % (k, call)] = v
        out["u0_%d" % call] = u
    np.savez_compressed(os.path.join(GOLDEN, "stages_%s.npz" % name), model=np.array(model), **out)
    return out


def main():
    os.makedirs(GOLDEN, exist_ok=True)
    for name in CONFIGS:
        out = generate(name)
        print("%-28s T=%3d  u0 of call 2: %s" % (name, out["traj_0"].shape[1], np.array2string(out["u0_2"], precision=6)))


if __name__ == "__main__":
    main()
