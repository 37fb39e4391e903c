"""Lanes per agent x batch size x steps per launch on the short-horizon BASELINE shapes (round 5).

For each shape (configs[0] Omni K5 T5, configs[1] SimpleCart K10 T20, yaml-as-shipped K10 T50) and each forced
EEA_OPT_AGENT_LANES in {64, 32, 16, 8} (where eligible) and each batch size: microseconds per pass (HIP events on the
launch streams, two agent groups on two streams as in bench.py), per 4096 agents, and the fraction of the fp64 vector
peak of the reference-formulation work W.  Output: one line per point + a JSON dump.

    python tools/pack_sweep.py [--spl 50] [--batches 4096,16384,32768] [--out gpurun_out/pack_sweep.json]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ergodic_exploration_amd import capi  # noqa: E402

MAP_BOUNDS = (-1.0, 11.0, -1.0, 5.0)
MEANS, SIGMAS = [[2.5, 2.5], [8.5, 2.5]], [[1.5, 1.5], [1.5, 1.5]]
PEAK = 78.6

SHAPES = [
    dict(name="configs[0]", model="omni", K=5, dt=0.1, horizon=0.5, means=[[2.5, 2.5]], sigmas=[[1.5, 1.5]]),
    dict(name="configs[1]", model="simple_cart", K=10, dt=0.1, horizon=2.0, means=MEANS, sigmas=SIGMAS),
    dict(name="yaml T=50", model="omni", K=10, dt=0.1, horizon=5.0, means=MEANS, sigmas=SIGMAS),
]


def measure(eng, B, T, spl, groups=2, target_s=0.15):
    rng = np.random.default_rng(777)
    b = MAP_BOUNDS
    poses = np.stack([rng.uniform(0.5, 11.5, B) + b[0], rng.uniform(0.5, 5.5, B) + b[2], rng.uniform(-np.pi, np.pi, B)], 1)
    d_pose = torch.as_tensor(poses).cuda()
    d_ut = torch.zeros((B, T, 3), dtype=torch.float64, device="cuda")
    d_u0 = torch.empty((B, 3), dtype=torch.float64, device="cuda")
    streams = [torch.cuda.Stream() for _ in range(groups)]
    cuts = [B * g // groups for g in range(groups + 1)]
    calls = [eng.prepared_batch(hi - lo, d_pose[lo:hi], d_ut[lo:hi], d_u0[lo:hi], stream=st.cuda_stream,
                                n_steps=None if spl == 1 else spl)
             for lo, hi, st in zip(cuts[:-1], cuts[1:], streams)]

    def run(n_calls):
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        ev0.record(streams[0])
        for st in streams[1:]:
            st.wait_event(ev0)
        for _ in range(n_calls):
            for call in calls:
                call()
        for st in streams[1:]:
            ej = torch.cuda.Event()
            ej.record(st)
            streams[0].wait_event(ej)
        ev1.record(streams[0])
        torch.cuda.synchronize()
        return ev0.elapsed_time(ev1) / (n_calls * spl)

    probe = run(max(1, 100 // spl))
    n = max(2, int(target_s / (probe * 1e-3) / spl))
    run(max(1, n // 4))
    return min(run(n), run(n))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--spl", default="50")
    ap.add_argument("--batches", default="4096,16384,32768")
    ap.add_argument("--lanes", default="64,32,16,8")
    ap.add_argument("--groups", type=int, default=2)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    rows = []
    for sh in SHAPES:
        if sh["model"] == "simple_cart":
            model, rdiag, lim = capi.MODEL_SIMPLE_CART, [1.0, 0.0, 2.0], np.array([1.0, 0.0, 2.0])
        else:
            model, rdiag, lim = capi.MODEL_OMNI, [1.0, 1.0, 2.0], np.array([1.0, 1.0, 2.0])
        eng = capi.Engine(capi.make_config(model, sh["dt"], sh["horizon"], 0.1, 1.0, sh["K"], np.diag(rdiag), -lim, lim))
        eng.set_target_gaussians(sh["means"], sh["sigmas"])
        eng.config_domain(MAP_BOUNDS)
        T, K = eng.T, sh["K"]
        flops = 2 * K * K * T + 4 * K * K * T + (4 * K + 140) * T
        for spl in [int(s) for s in args.spl.split(",")]:
            for B in [int(s) for s in args.batches.split(",")]:
                for L in [int(s) for s in args.lanes.split(",")]:
                    capi.set_option(capi.OPT_AGENT_LANES, L)
                    got = eng.agent_lanes(B // args.groups)
                    if got != L:
                        continue
                    ms = measure(eng, B, T, spl, groups=args.groups)
                    us4096 = ms * 1e3 * 4096 / B
                    frac = flops * B / (ms * 1e-3) / 1e12 / PEAK
                    row = dict(shape=sh["name"], K=K, T=T, lanes=L, agents=B, steps_per_launch=spl, us_per_pass=ms * 1e3,
                               us_per_4096=us4096, frac=frac)
                    rows.append(row)
                    print("%-11s K%-2d T%-3d L=%-2d B=%-6d spl=%-3d  %8.2f us/pass  %6.2f us/4096  frac %.3f" %
                          (sh["name"], K, T, L, B, spl, ms * 1e3, us4096, frac), flush=True)
        capi.set_option(capi.OPT_AGENT_LANES, 0)
        for B in [int(s) for s in args.batches.split(",")]:
            print("  automatic choice at B=%d (per group of %d): %d lanes" % (B, B // args.groups, eng.agent_lanes(B // args.groups)))
        eng.close()
    if args.out:
        os.makedirs(os.path.dirname(args.out), exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
