#!/usr/bin/env python3
"""Headline benchmark: receding-horizon optimisations/s at K = 10x10, T = 200, agent batch.

One "step" = one batched `ErgodicControl::control` pass (eea_control_batch) over the rank's
agents, each agent one complete receding-horizon optimisation.  Weak scaling: every rank (one
process per GPU) owns `--agents` independent agents; after each step the per-agent c_k are
all-gathered over RCCL (overlapped with the next step on a separate stream).

  python bench.py --gpus 1 --steps 20 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
      --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MAP_BOUNDS = (-1.0, 11.0, -1.0, 5.0)
MEANS = [[2.5, 2.5], [8.5, 2.5]]
SIGMAS = [[1.5, 1.5], [1.5, 1.5]]
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s
VALU_F64_PEAK_TF = 78.6     # fp64 vector peak
VALU_F32_PEAK_TF = 157.3


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--agents", type=int, default=4096, help="agents per GPU")
    ap.add_argument("--model", default="simple_cart", choices=["simple_cart", "omni"])
    ap.add_argument("--num-basis", type=int, default=10)
    ap.add_argument("--horizon", type=float, default=20.0)
    ap.add_argument("--dt", type=float, default=0.1)
    ap.add_argument("--n-mem", type=int, default=0)
    ap.add_argument("--precision", default="f64", choices=["f64", "f32"])
    ap.add_argument("--no-gather", action="store_true", help="skip the c_k all-gather (N > 1)")
    ap.add_argument("--force-gather", action="store_true",
                    help="run the c_k all-gather even with one rank (exercises the RCCL path on 1 GPU)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline budget per leg; 0 = skip")
    ap.add_argument("--latency", action="store_true", help="(default on one GPU) also time the B = 1 dependent-call mode")
    ap.add_argument("--no-latency", action="store_true", help="skip the B = 1 dependent-call leg")
    ap.add_argument("--agent-groups", type=int, default=1,
                    help="split the rank's agents into this many groups, each launched on its own HIP stream: the "
                         "drain of one group's launch overlaps the fill of another's (agents are independent)")
    return ap.parse_args()


def cpu_baseline(args, T, seconds):
    """Oracle (literal CPU restatement, kind 'port') timed on this host: 1 thread and all cores."""
    from oracle import pyoracle as po
    lim = np.array([1.0, 0.0, 2.0]) if args.model == "simple_cart" else np.array([1.0, 1.0, 2.0])
    Rinv = np.diag([1.0, 0.0, 2.0]) if args.model == "simple_cart" else np.diag([1.0, 1.0, 2.0])
    model = po.MODEL_SIMPLE_CART if args.model == "simple_cart" else po.MODEL_OMNI
    cfg = po.make_config(model, args.dt, args.horizon, 0.1, 1.0, args.num_basis, Rinv, -lim, lim)
    rng = np.random.default_rng(12345)
    ncores = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:  # a cgroup CPU quota (e.g. "1600000 100000" = 16 CPUs) caps the useful thread count
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            ncores = max(1, min(ncores, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass

    def poses(n):
        return np.stack([rng.uniform(-0.5, 10.5, n), rng.uniform(-0.5, 4.5, n), rng.uniform(-np.pi, np.pi, n)], 1)

    # calibrate each leg on a tiny sample, then size its timed sample to the budget, so the
    # default run stays bounded whatever the host's core count / CPU quota is
    calls = 10
    sec, _ = po.bench_control(cfg, MEANS, SIGMAS, MAP_BOUNDS, poses(4), 5, 1)
    rate1 = 20.0 / max(sec, 1e-6)
    n1 = max(4, int(seconds * rate1 / calls))
    sec1, _ = po.bench_control(cfg, MEANS, SIGMAS, MAP_BOUNDS, poses(n1), calls, 1)
    one = {"value": n1 * calls / sec1, "unit": "optimisations/s", "cores": 1, "kind": "port",
           "sample": "%d agents x %d control() calls, oracle/ergodic_oracle.c gcc -O2, 1 thread, %.1f s"
                     % (n1, calls, sec1)}
    secc, _ = po.bench_control(cfg, MEANS, SIGMAS, MAP_BOUNDS, poses(2 * ncores), 3, ncores)
    raten = 6.0 * ncores / max(secc, 1e-6)
    nall = max(ncores, int(seconds * raten / calls))
    secn, _ = po.bench_control(cfg, MEANS, SIGMAS, MAP_BOUNDS, poses(nall), calls, ncores)
    allc = {"value": nall * calls / secn, "unit": "optimisations/s", "cores": ncores, "kind": "port",
            "sample": "%d agents x %d control() calls, one agent per thread, %d threads, %.1f s"
                      % (nall, calls, ncores, secn)}
    return one, allc


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    from ergodic_exploration_amd import capi

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    # plumbing test on a single-GPU box: EEA_DIST_BACKEND=gloo with --no-gather runs several ranks on one
    # device (RCCL refuses two ranks per GPU); on a real node every rank has its own device
    backend = os.environ.get("EEA_DIST_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.force_gather
    if use_dist:
        # RCCL writes its banner / warnings to stdout; keep stdout for the ONE JSON line
        os.environ["NCCL_DEBUG"] = os.environ.get("EEA_NCCL_DEBUG", "WARN")
        os.environ.setdefault("NCCL_DEBUG_FILE", "/tmp/eea_rccl_%h_%p.log")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    f32 = args.precision == "f32"
    tdt = torch.float32 if f32 else torch.float64
    rs = 4 if f32 else 8
    if args.model == "simple_cart":
        model, rdiag, lim = capi.MODEL_SIMPLE_CART, [1.0, 0.0, 2.0], np.array([1.0, 0.0, 2.0])
    else:
        model, rdiag, lim = capi.MODEL_OMNI, [1.0, 1.0, 2.0], np.array([1.0, 1.0, 2.0])
    eng = capi.Engine(capi.make_config(model, args.dt, args.horizon, 0.1, 1.0, args.num_basis,
                                       np.diag(rdiag), -lim, lim,
                                       precision=capi.PREC_F32 if f32 else capi.PREC_F64,
                                       device=local_rank))
    eng.set_target_gaussians(MEANS, SIGMAS)
    eng.config_domain(MAP_BOUNDS)
    T, K2, B = eng.T, eng.K2, args.agents

    # synthetic inputs: random poses, zero warm start
    rng = np.random.default_rng(12345 + rank)
    poses = np.stack([rng.uniform(0.5, 11.5, B) - 1.0, rng.uniform(0.5, 5.5, B) - 1.0,
                      rng.uniform(-np.pi, np.pi, B)], 1)
    d_pose = torch.as_tensor(poses, dtype=tdt).cuda()
    d_ut = torch.zeros((B, T, 3), dtype=tdt, device="cuda")
    d_u0 = torch.empty((B, 3), dtype=tdt, device="cuda")
    d_ck = [torch.empty((B, K2), dtype=tdt, device="cuda") for _ in range(2)]
    d_mem = d_nmem = None
    if args.n_mem:
        mem = np.stack([rng.uniform(0.5, 11.5, B * args.n_mem) - 1.0, rng.uniform(0.5, 5.5, B * args.n_mem) - 1.0,
                        rng.uniform(-np.pi, np.pi, B * args.n_mem)], 1).reshape(B, args.n_mem, 3)
        d_mem = torch.as_tensor(mem, dtype=tdt).cuda()
        d_nmem = torch.full((B,), args.n_mem, dtype=torch.int32, device="cuda")
    gather = (world > 1 or args.force_gather) and not args.no_gather
    d_all = [torch.empty((world * B, K2), dtype=tdt, device="cuda") for _ in range(2)] if gather else None

    compute = torch.cuda.Stream()
    torch.cuda.set_stream(compute)  # once, not per step: the step loop must out-run a 60 us kernel
    cstream = compute.cuda_stream
    works = [None, None]
    all_gather = dist.all_gather_into_tensor if gather else None
    # agent groups: contiguous slices of the batch, group 0 on the compute stream
    G = max(1, min(args.agent_groups, B))
    bounds = [(g * B) // G for g in range(G + 1)]
    gstreams = [compute] + [torch.cuda.Stream() for _ in range(G - 1)]
    gevents = [torch.cuda.Event() for _ in range(G)]

    def sl(t, g):
        return None if t is None else t[bounds[g]:bounds[g + 1]]

    # per-group argument tuples, built once (tensor slicing costs host time on every step otherwise)
    gargs = [[dict(B=bounds[g + 1] - bounds[g], pose=sl(d_pose, g), ut=sl(d_ut, g), u0=sl(d_u0, g),
                   mem_cols=sl(d_mem, g), n_mem=sl(d_nmem, g), ck=sl(d_ck[slot], g),
                   stream=gstreams[g].cuda_stream) for g in range(G)] for slot in range(2)]

    def step(i):
        slot = i & 1
        if gather and works[slot] is not None:
            works[slot].wait()  # the gather that read this slot two steps ago has finished
        for a in gargs[slot]:
            eng.control_batch(a["B"], a["pose"], a["ut"], a["u0"], mem_cols=a["mem_cols"], n_mem=a["n_mem"],
                              mem_stride=args.n_mem, ck=a["ck"], stream=a["stream"])
        if gather:
            # RCCL all-gather of the per-agent c_k over xGMI; runs on the process group's
            # stream and overlaps with the next step's kernel
            for g in range(1, G):  # the gather reads every group's c_k
                gevents[g].record(gstreams[g])
                compute.wait_event(gevents[g])
            works[slot] = all_gather(d_all[slot], d_ck[slot], async_op=True)

    def drain():
        for w in works:
            if w is not None:
                w.wait()
        torch.cuda.synchronize()  # all streams of the device

    for i in range(args.warmup):
        step(i)
    drain()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(compute)
    for i in range(args.steps):
        step(args.warmup + i)
    ev1.record(compute)
    enqueue_s = time.perf_counter() - t0  # host time to enqueue all steps (must stay below the device time)
    drain()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / args.steps  # HIP events on the kernel's own stream

    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        k = torch.tensor([kernel_ms], dtype=torch.float64, device="cuda")
        dist.all_reduce(k, op=dist.ReduceOp.MAX)
        kernel_ms = float(k.item())

    if rank == 0:
        N = T + args.n_mem
        K = args.num_basis
        total_opts = world * B * args.steps
        value = total_opts / elapsed
        # algorithmic HBM bytes per optimisation (DESIGN.md "Control kernel roofline"):
        # pose in + ut in + ut out + u0 out + c_k out (+ memory columns in)
        bytes_per_opt = rs * (3 + 3 * T + 3 * T + 3 + K2 + 3 * args.n_mem)
        flops_per_opt = 2 * K * K * N + 4 * K * K * T + (4 * K + 140) * T  # SURVEY.md 8(d) "W"
        launch_s = kernel_ms * 1e-3
        # per launch: with agent groups, the launches of group 0 (timed by the events on its stream)
        Bl = bounds[1] - bounds[0]
        hbm_gbs = bytes_per_opt * Bl / launch_s / 1e9
        tflops = flops_per_opt * Bl / launch_s / 1e12
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "r01_control_pmc.json")
        if os.path.exists(pmc):
            try:
                with open(pmc) as f:
                    rec = json.load(f)
                if rec.get("agents") == Bl and rec.get("T") == T and rec.get("K") == K and rec.get("precision") == args.precision:
                    traffic = rec.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        host_us = 1e6 * enqueue_s / args.steps
        out = {
            "metric": "receding-horizon optimisations/sec at K=10x10, T=200; 1/2/4/8-GPU agent-batch",
            "value": value, "unit": "optimisations/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic",
            "config": {"workload": "BASELINE configs[3]: %d-agent batch per GPU, %s, two-Gaussian 12x6 m map, "
                                   "K=%dx%d, T=%d steps (dt %.3g, horizon %.3g), n_mem=%d, c_k all-gather %s"
                                   % (B, args.model, K, K, T, args.dt, args.horizon, args.n_mem,
                                      "on" if gather else "off (1 GPU)"),
                       "agents_per_gpu": B, "num_basis": K, "horizon_steps": T, "kinematics": args.model,
                       "agent_groups": G,
                       "parallelism": "agent-batch x%d" % world},
            "host_enqueue_us_per_step": host_us,
            "roofline": {"bound": "hbm", "kernel": "control_kernel", "achieved": hbm_gbs, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": hbm_gbs / HBM_PEAK_GBS, "traffic": traffic,
                         "bytes_per_launch": bytes_per_opt * Bl, "launch_ms": kernel_ms,
                         "agents_per_launch": Bl, "concurrent_launches": G},
            "roofline_valu": {"bound": "valu-%s" % args.precision, "kernel": "control_kernel",
                              "achieved": tflops, "peak": VALU_F32_PEAK_TF if f32 else VALU_F64_PEAK_TF,
                              "unit": "TFLOP/s",
                              "frac": tflops / (VALU_F32_PEAK_TF if f32 else VALU_F64_PEAK_TF),
                              "flops_per_launch": flops_per_opt * Bl,
                              "note": "the control kernel is vector-ALU/transcendental bound, not HBM bound "
                                      "(SURVEY.md 8(d)); W = 2K^2N + 4K^2T + (4K+140)T flop per optimisation"},
        }
        if world == 1 and args.cpu_seconds > 0:
            one, allc = cpu_baseline(args, T, args.cpu_seconds)
            out["cpu_baseline"] = one
            out["cpu_baseline_all_cores"] = allc
        if world == 1 and not args.no_latency:
            x = poses[0].astype(np.float64)
            for _ in range(20):
                eng.control(MAP_BOUNDS, x)
            n = 500
            t0 = time.perf_counter()
            for _ in range(n):
                eng.control(MAP_BOUNDS, x)
            lat = (time.perf_counter() - t0) / n
            out["latency_mode"] = {"value": 1.0 / lat, "unit": "optimisations/s", "us_per_call": 1e6 * lat,
                                   "note": "B = 1, dependent eea_control calls incl. host round trip"}
        result_line = json.dumps(out)
    eng.close()
    if use_dist:
        dist.destroy_process_group()
    if rank == 0:
        # last thing on stdout: the ONE JSON line, written in one piece after every C-level
        # stdio buffer (collective library banners) has been flushed
        import ctypes
        sys.stdout.flush()
        try:
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        os.write(1, (result_line + "\n").encode())


if __name__ == "__main__":
    main()
