"""One short-horizon shape, one group size, one batch: a few launches (for rocprofv3 --kernel-trace / --pmc).

    rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU ... -- python3 tools/pack_point.py --shape 1 --lanes 8 --agents 24576
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ergodic_exploration_amd import capi  # noqa: E402
from pack_sweep import MAP_BOUNDS, SHAPES  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", type=int, default=1)
    ap.add_argument("--lanes", type=int, default=8)
    ap.add_argument("--agents", type=int, default=24576)
    ap.add_argument("--spl", type=int, default=1)
    ap.add_argument("--launches", type=int, default=10)
    ap.add_argument("--workgroup", action="store_true", help="the workgroup-per-agent kernel (EEA_OPT_CONTROL_KERNEL = 1)")
    ap.add_argument("--steps", type=int, default=0, help="horizon steps (overrides the shape's horizon; dt = 0.125)")
    a = ap.parse_args()
    sh = dict(SHAPES[a.shape])
    if a.steps:
        sh["dt"], sh["horizon"] = 0.125, 0.125 * a.steps
    if sh["model"] == "simple_cart":
        model, rdiag, lim = capi.MODEL_SIMPLE_CART, [1.0, 0.0, 2.0], np.array([1.0, 0.0, 2.0])
    else:
        model, rdiag, lim = capi.MODEL_OMNI, [1.0, 1.0, 2.0], np.array([1.0, 1.0, 2.0])
    eng = capi.Engine(capi.make_config(model, sh["dt"], sh["horizon"], 0.1, 1.0, sh["K"], np.diag(rdiag), -lim, lim))
    eng.set_target_gaussians(sh["means"], sh["sigmas"])
    eng.config_domain(MAP_BOUNDS)
    capi.set_option(capi.OPT_AGENT_LANES, a.lanes)
    B, T = a.agents, eng.T
    if a.workgroup:
        capi.set_option(capi.OPT_CONTROL_KERNEL, 1)
    else:
        assert eng.agent_lanes(B) == a.lanes, eng.agent_lanes(B)
    rng = np.random.default_rng(777)
    b = MAP_BOUNDS
    poses = np.stack([rng.uniform(0.5, 11.5, B) + b[0], rng.uniform(0.5, 5.5, B) + b[2], rng.uniform(-np.pi, np.pi, B)], 1)
    d_pose = torch.as_tensor(poses).cuda()
    d_ut = torch.zeros((B, T, 3), dtype=torch.float64, device="cuda")
    d_u0 = torch.empty((B, 3), dtype=torch.float64, device="cuda")
    call = eng.prepared_batch(B, d_pose, d_ut, d_u0, n_steps=None if a.spl == 1 else a.spl)
    for _ in range(a.launches):
        call()
    torch.cuda.synchronize()
    print("%s K=%d T=%d lanes=%d agents=%d spl=%d launches=%d" % (sh["name"], sh["K"], T, a.lanes, B, a.spl, a.launches))
    eng.close()


if __name__ == "__main__":
    main()
