#!/usr/bin/env python3
"""Headline benchmark: receding-horizon optimisations/s at K = 10x10, T = 200, agent batch.

One "pass" = one batched `ErgodicControl::control` (eea_control_batch) over the rank's agents, each agent one
complete receding-horizon optimisation.  One "step" = `--passes-per-step` consecutive passes (the receding
horizon: every pass starts from the controls the previous one left), so that the timed region of the
driver's `--steps 20` lasts tens of milliseconds instead of one.

Weak scaling: every rank (one process per GPU) owns `--agents` independent agents; the control computation
has no data-path collective, and `value` is that pure agent shard.  The exchange steps the agent batch can
run on top are timed as separate legs and reported under "exchange": the consensus c_k (one all-reduce of
K^2 + 1 reals per pass, consumed by the next passes through eea_batch_io::d_ck_shared) and the all-gather of
every agent's c_k that north_star names.

  python bench.py                                  # 1 GPU
  python bench.py --gpus N --steps K --warmup W    # starts its N ranks itself (one child process per GPU)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
      --master-port P bench.py --gpus N --steps K --warmup W      # what the driver runs for N > 1
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import bench_line  # noqa: E402 -- the compact driver line + bench_detail.json

MAP_BOUNDS = (-1.0, 11.0, -1.0, 5.0)
MEANS = [[2.5, 2.5], [8.5, 2.5]]
SIGMAS = [[1.5, 1.5], [1.5, 1.5]]
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s
VALU_F64_PEAK_TF = 78.6     # fp64 vector peak
VALU_F32_PEAK_TF = 157.3
SPINUP_PASSES = int(os.environ.get("EEA_BENCH_SPINUP_PASSES", "1000"))  # untimed passes before the warm-up steps (clock ramp)
PARITY_TOL = {"f64": {"ck_phik": 1e-11, "traj_rho_u": 1e-9}, "f32": {"u": 1e-4, "rho": 5e-4}}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--passes-per-step", type=int, default=16000,
                    help="consecutive control passes per timed step (receding horizon); the default keeps the GPU "
                         "busy for >= 2 s over the driver's --steps 20")
    ap.add_argument("--exchange-passes-per-step", type=int, default=400,
                    help="passes per step of the secondary legs (exchange, grid tile)")
    ap.add_argument("--control-kernel", default="auto", choices=["auto", "workgroup"],
                    help="eea_set_option(EEA_OPT_CONTROL_KERNEL): workgroup = the workgroup-per-agent kernel for every call")
    ap.add_argument("--workgroup-threads", type=int, default=0, choices=[0, 64, 128, 256],
                    help="eea_set_option(EEA_OPT_WORKGROUP_THREADS)")
    ap.add_argument("--no-grid-tile", action="store_true", help="skip the grid-tile leg (BASELINE config 5 shard)")
    ap.add_argument("--agents", type=int, default=4096, help="agents per GPU")
    ap.add_argument("--model", default="simple_cart", choices=["simple_cart", "omni"])
    ap.add_argument("--num-basis", type=int, default=10)
    ap.add_argument("--horizon", type=float, default=20.0)
    ap.add_argument("--dt", type=float, default=0.1)
    ap.add_argument("--n-mem", type=int, default=0)
    ap.add_argument("--precision", default="f64", choices=["f64", "f32"])
    ap.add_argument("--no-exchange", "--no-gather", dest="no_exchange", action="store_true",
                    help="skip the exchange legs (consensus all-reduce, c_k all-gather)")
    ap.add_argument("--force-exchange", "--force-gather", dest="force_exchange", action="store_true",
                    help="run the all-gather leg even with one rank (single-rank RCCL communicator)")
    ap.add_argument("--consensus-lags", default="1,2,4",
                    help="the consensus leg is timed once per lag: pass i consumes the c_bar of pass i - lag.  Lag 1 is the "
                         "previous step's consensus (decentralised ergodic control); the exchange is device-bound "
                         "(eea_comm_records_exchange_bound): no host wait, no stream wait, no host thread at any lag")
    ap.add_argument("--consensus-buffers", type=int, default=8, choices=range(3, 9),
                    help="record / sum buffers (and exchange slots) the consensus leg rotates through")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline budget per leg; 0 = skip")
    ap.add_argument("--no-latency", action="store_true", help="skip the B = 1 dependent-call leg")
    ap.add_argument("--no-phik", action="store_true", help="skip the phi_k legs (roofline_phik)")
    ap.add_argument("--no-single-launch", action="store_true",
                    help="skip the short one-launch-per-pass leg behind the headline leg (counter-collection runs: every "
                         "control dispatch of the run then has the headline's shape)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the legs of the other single-GPU BASELINE configs (configs[1], configs[2] fp32 + fp64)")
    ap.add_argument("--phik-grid", type=int, default=16384, help="side of the square fp64 grid of the phi_k leg")
    ap.add_argument("--steps-per-launch", type=int, default=50,
                    help="receding-horizon steps (passes) per launch of the shard leg (eea_control_batch_steps: the agent's "
                         "wavefront carries on with its own stored controls; must divide --passes-per-step)")
    ap.add_argument("--agent-groups", type=int, default=2,
                    help="split the rank's agents into this many contiguous groups, each stepped by its own "
                         "eea_control_batch call on its own HIP stream: agents are independent, and the head of one "
                         "group's launch (loading the controls) overlaps the body of the other's (measured: "
                         "profiles/r02_ablation.txt); 1 = one launch per pass")
    return ap.parse_args()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(args):
    """`python bench.py --gpus N` with no launcher around it: this process has made NO GPU call (torch is not
    even imported); it starts the N ranks as children of torch.distributed.run, forwards their output (rank 0
    prints the ONE JSON line) and exits with their return code."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def host_info():
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    ncores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    usable = ncores
    try:  # a cgroup CPU quota (e.g. "1600000 100000" = 16 CPUs) caps the useful thread count
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            usable = max(1, min(ncores, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    cflags = "unknown"
    try:
        with open(os.path.join(ROOT, "oracle", "Makefile")) as f:
            for line in f:
                if line.startswith("CFLAGS"):
                    cflags = "gcc " + line.split("=", 1)[1].strip()
    except OSError:
        pass
    return {"cpu_model": model, "nproc": os.cpu_count() or 1, "usable_threads": usable, "cflags": cflags}


def cpu_baseline(args, seconds):
    """Oracle (literal CPU restatement, kind 'port') timed on this host: 1 thread and all usable cores."""
    import numpy as np
    from oracle import pyoracle as po
    lim = np.array([1.0, 0.0, 2.0]) if args.model == "simple_cart" else np.array([1.0, 1.0, 2.0])
    Rinv = np.diag([1.0, 0.0, 2.0]) if args.model == "simple_cart" else np.diag([1.0, 1.0, 2.0])
    model = po.MODEL_SIMPLE_CART if args.model == "simple_cart" else po.MODEL_OMNI
    cfg = po.make_config(model, args.dt, args.horizon, 0.1, 1.0, args.num_basis, Rinv, -lim, lim)
    rng = np.random.default_rng(12345)
    hi = host_info()
    ncores = hi["usable_threads"]

    def poses(n):
        return np.stack([rng.uniform(-0.5, 10.5, n), rng.uniform(-0.5, 4.5, n), rng.uniform(-np.pi, np.pi, n)], 1)

    # calibrate each leg on a tiny sample, then size its timed sample to the budget, so the
    # default run stays bounded whatever the host's core count / CPU quota is
    calls = 10
    sec, _ = po.bench_control(cfg, MEANS, SIGMAS, MAP_BOUNDS, poses(4), 5, 1)
    rate1 = 20.0 / max(sec, 1e-6)
    n1 = max(4, int(seconds * rate1 / calls))
    sec1, _ = po.bench_control(cfg, MEANS, SIGMAS, MAP_BOUNDS, poses(n1), calls, 1)
    common = {"unit": "optimisations/s", "kind": "port", "cpu_model": hi["cpu_model"], "nproc": hi["nproc"],
              "cflags": hi["cflags"]}
    one = dict(common, value=n1 * calls / sec1, cores=1,
               sample="%d agents x %d control() calls, oracle/ergodic_oracle.c, 1 thread, %.1f s" % (n1, calls, sec1))
    secc, _ = po.bench_control(cfg, MEANS, SIGMAS, MAP_BOUNDS, poses(2 * ncores), 3, ncores)
    raten = 6.0 * ncores / max(secc, 1e-6)
    nall = max(ncores, int(seconds * raten / calls))
    secn, _ = po.bench_control(cfg, MEANS, SIGMAS, MAP_BOUNDS, poses(nall), calls, ncores)
    allc = dict(common, value=nall * calls / secn, cores=ncores,
                sample="%d agents x %d control() calls, one agent per thread, %d threads, %.1f s"
                       % (nall, calls, ncores, secn))
    return one, allc


def phik_legs(args, torch, capi, np):
    """The second kernel of the path (two kernels, two bounds): Basis::spatialCoeff streaming a
    target grid larger than the Infinity Cache against the HBM roofline, and the wall time of a whole
    configTarget rebuild (eea_config_domain) at the BASELINE grids."""
    out = {}
    n, K = args.phik_grid, args.num_basis
    eng = capi.Engine(capi.make_config(capi.MODEL_OMNI, 0.1, 2.0, 0.1, 1.0, K, np.eye(3), [-1] * 3, [1] * 3))
    phi = torch.rand((n * n,), dtype=torch.float64, device="cuda")
    part = torch.empty((K * K,), dtype=torch.float64, device="cuda")
    lx = ly = (n - 1) * 0.1
    stream = torch.cuda.current_stream()
    for _ in range(3):  # the first call builds the axis tables; later calls only stream the grid
        eng.spatial_coeff_rows(n, n, 0, n, phi, lx, ly, part, stream=stream.cuda_stream)
    torch.cuda.synchronize()
    reps = 10
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record(stream)
    for _ in range(reps):
        eng.spatial_coeff_rows(n, n, 0, n, phi, lx, ly, part, stream=stream.cuda_stream)
    ev1.record(stream)
    torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / reps
    nbytes = n * n * 8
    gbs = nbytes / (ms * 1e-3) / 1e9
    traffic, traffic_source = None, "profiles/ (rocprofv3 --pmc passes, not collected in-run)"
    try:
        pname = "r05_phik_pmc.json" if os.path.exists(os.path.join(ROOT, "profiles", "r05_phik_pmc.json")) else "r03_phik_pmc.json"
        with open(os.path.join(ROOT, "profiles", pname)) as f:
            rec = json.load(f)
        if rec.get("grid") == n and rec.get("K") == K and rec.get("precision") == "f64":
            traffic = rec["hbm_read_bytes_x2_corrected"]
            traffic_source = ("profiles/%s (separate rocprofv3 --pmc FETCH_SIZE pass of this workload, x2 "
                              "gfx950 correction; NOT measured in this run)" % pname)
    except Exception:
        pass
    out["roofline_phik"] = {"bound": "hbm", "kernel": "spatial_stream_kernel (+ sum_partials_kernel)",
                            "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                            "traffic": traffic, "traffic_source": traffic_source,
                            "bytes_per_launch": nbytes, "launch_ms": ms,
                            "workload": "Basis::spatialCoeff, %dx%d fp64 target grid (%.2f GB) resident in HBM, K=%d"
                                        % (n, n, nbytes / 1e9, K),
                            "note": "a SYNTHETIC grid sized to expose the HBM roofline of the streaming kernel.  The grids BASELINE "
                                    "names (121x61, 256x256, 1024x1024: 59 KB - 8 MB) are launch-latency bound -- one workgroup from "
                                    "the per-axis factors, 7 - 17 us of device time whatever the bytes (config_domain_rebuild below: "
                                    "the 1024x1024 rebuild moves 1 MB in ~17 us = 0.8 %% of HBM)"}
    eng.close()
    del phi, part
    torch.cuda.empty_cache()
    # whole rebuild through the reference's entry (configTarget with a changed extent), Gaussian target: the synchronous
    # form (returns when phi_k is on the device) and the enqueue-only form (returns when the launches are on the
    # stream; the next control call on that stream is ordered behind them)
    rebuild = []
    st = torch.cuda.Stream()
    for Kc, lxc, lyc in ((10, 12.0, 6.0), (20, 25.5, 25.5), (30, 102.3, 102.3)):
        e2 = capi.Engine(capi.make_config(capi.MODEL_OMNI, 0.1, 2.0, 0.1, 1.0, Kc, np.eye(3), [-1] * 3, [1] * 3))
        e2.set_target_gaussians(MEANS, SIGMAS)
        for i in range(4):
            e2.config_domain((0.0, lxc + 0.1 * (i % 2), 0.0, lyc))
        reps = 50
        t0 = time.perf_counter()
        for i in range(reps):
            e2.config_domain((0.0, lxc + 0.1 * (i % 2), 0.0, lyc))  # the extent changes on every call
        us = 1e6 * (time.perf_counter() - t0) / reps
        torch.cuda.synchronize()
        # enqueue-only: host time per call with the device keeping up (a stream synchronisation every 10 calls, timed
        # apart), and the device time per rebuild from HIP events around a back-to-back run
        enq = 0.0
        for i in range(reps):
            t0 = time.perf_counter()
            e2.config_domain_async((0.0, lxc + 0.1 * (i % 2), 0.0, lyc), stream=st.cuda_stream)
            enq += time.perf_counter() - t0
            if i % 10 == 9:
                st.synchronize()
        st.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record(st)
        for i in range(reps):
            e2.config_domain_async((0.0, lxc + 0.1 * (i % 2), 0.0, lyc), stream=st.cuda_stream)
        ev1.record(st)
        st.synchronize()
        rebuild.append({"K": Kc, "grid": "%dx%d" % (round(lxc / 0.1) + 1, round(lyc / 0.1) + 1), "wall_us": us,
                        "enqueue_only_wall_us": 1e6 * enq / reps, "device_us_back_to_back": 1e3 * ev0.elapsed_time(ev1) / reps})
        e2.close()
    out["config_domain_rebuild"] = {"note": "eea_config_domain with a changed extent (Target::fill + normalisation + spatialCoeff "
                                            "on the device): wall_us = synchronous form (one host wait); enqueue_only_wall_us = "
                                            "eea_config_domain_async (the caller's thread is free again; the next control call on "
                                            "the stream is ordered behind the rebuild); device_us = stream time per rebuild",
                                    "cases": rebuild}
    return out


def _profiled_short_horizons():
    """profiles/r06_pack_profile.json: {(config name, agents, steps per launch): record}"""
    try:
        with open(os.path.join(ROOT, "profiles", "r06_pack_profile.json")) as f:
            return {(r["config"], r["agents"], r["steps_per_launch"]): r for r in json.load(f)["cases"]}
    except Exception:  # noqa: BLE001
        return {}


def other_config_legs(args, torch, capi, np, spl, only=None, spinup_s=0.0):
    """The other single-GPU BASELINE configurations, 4096 agents each, ~0.3 s timed, the same launch form as the headline
    (two agent groups, `spl` receding-horizon steps per launch): configs[1] (SimpleCart, K = 10, horizon 2 s @ 0.1: T = 20,
    fp64) and configs[2] (Omni, K = 20, horizon 5 s @ 0.02: T = 250, 256 x 256 target grid, fp32 -- and its fp64 twin).
    Per leg: ms per pass, roofline fraction against the dtype's own vector peak, and the configTarget rebuild of that grid
    (config/explore_omni.yaml:49-56 for the parameter names)."""
    import time as _t
    cases = [
        # short horizons: several agents share a wavefront (csrc/control_pack_impl.hpp) -- at the headline's 4096 agents a
        # pass is latency-bound (one or two wavefronts per SIMD), so each shape is also timed at the batch that fills the
        # chip with resident wavefronts of the engine's choice of lanes per agent (4 x 1024 x 64 / lanes: four wavefronts per SIMD
        # since round 6, K = 10 too)
        dict(name="configs[0]", model="omni", K=5, dt=0.1, horizon=0.5, prec="f64", bounds=MAP_BOUNDS,
             means=[[2.5, 2.5]], sigmas=[[1.5, 1.5]]),
        dict(name="configs[0], chip-filling batch", model="omni", K=5, dt=0.1, horizon=0.5, prec="f64", bounds=MAP_BOUNDS,
             means=[[2.5, 2.5]], sigmas=[[1.5, 1.5]], agents=32768),
        dict(name="configs[1]", model="simple_cart", K=10, dt=0.1, horizon=2.0, prec="f64", bounds=MAP_BOUNDS,
             means=MEANS, sigmas=SIGMAS),
        dict(name="configs[1], chip-filling batch", model="simple_cart", K=10, dt=0.1, horizon=2.0, prec="f64", bounds=MAP_BOUNDS,
             means=MEANS, sigmas=SIGMAS, agents=32768),
        dict(name="explore_omni.yaml as shipped (K = 10, T = 50)", model="omni", K=10, dt=0.1, horizon=5.0, prec="f64",
             bounds=MAP_BOUNDS, means=MEANS, sigmas=SIGMAS),
        dict(name="explore_omni.yaml as shipped, chip-filling batch", model="omni", K=10, dt=0.1, horizon=5.0, prec="f64",
             bounds=MAP_BOUNDS, means=MEANS, sigmas=SIGMAS, agents=16384),
        dict(name="configs[2]", model="omni", K=20, dt=0.02, horizon=5.0, prec="f32", bounds=(0.0, 25.5, 0.0, 25.5),
             means=[[6.0, 6.0], [19.0, 12.0]], sigmas=[[3.0, 3.0], [3.0, 3.0]]),
        dict(name="configs[2] fp64 twin", model="omni", K=20, dt=0.02, horizon=5.0, prec="f64", bounds=(0.0, 25.5, 0.0, 25.5),
             means=[[6.0, 6.0], [19.0, 12.0]], sigmas=[[3.0, 3.0], [3.0, 3.0]]),
        # SURVEY.md 8(d) cfg 4's other halves: the Omni model, and a full replay-memory batch (n_mem = 100 sampled past states)
        dict(name="configs[3] with the Omni model", model="omni", K=10, dt=0.1, horizon=20.0, prec="f64", bounds=MAP_BOUNDS,
             means=MEANS, sigmas=SIGMAS),
        dict(name="configs[3] with n_mem = 100", model="simple_cart", K=10, dt=0.1, horizon=20.0, prec="f64", bounds=MAP_BOUNDS,
             means=MEANS, sigmas=SIGMAS, n_mem=100),
    ]
    res = []
    prof = _profiled_short_horizons()
    for c in cases:
        if only is not None and c["name"] not in only:
            continue
        B = c.get("agents", args.agents)
        f32 = c["prec"] == "f32"
        tdt = torch.float32 if f32 else torch.float64
        if c["model"] == "simple_cart":
            model, rdiag, lim = capi.MODEL_SIMPLE_CART, [1.0, 0.0, 2.0], np.array([1.0, 0.0, 2.0])
        else:
            model, rdiag, lim = capi.MODEL_OMNI, [1.0, 1.0, 2.0], np.array([1.0, 1.0, 2.0])
        eng = capi.Engine(capi.make_config(model, c["dt"], c["horizon"], 0.1, 1.0, c["K"], np.diag(rdiag), -lim, lim,
                                           precision=capi.PREC_F32 if f32 else capi.PREC_F64))
        eng.set_target_gaussians(c["means"], c["sigmas"])
        eng.config_domain(c["bounds"])
        T, K = eng.T, c["K"]
        b = c["bounds"]
        rng = np.random.default_rng(777)
        poses = np.stack([rng.uniform(0.5, b[1] - b[0] - 0.5, B) + b[0], rng.uniform(0.5, b[3] - b[2] - 0.5, B) + b[2],
                          rng.uniform(-np.pi, np.pi, B)], 1)
        d_pose = torch.as_tensor(poses, dtype=tdt).cuda()
        d_ut = torch.zeros((B, T, 3), dtype=tdt, device="cuda")
        d_u0 = torch.empty((B, 3), dtype=tdt, device="cuda")
        n_mem = c.get("n_mem", 0)
        d_mem = d_nmem = None
        if n_mem:
            mem = np.stack([rng.uniform(0.5, b[1] - b[0] - 0.5, B * n_mem) + b[0],
                            rng.uniform(0.5, b[3] - b[2] - 0.5, B * n_mem) + b[2],
                            rng.uniform(-np.pi, np.pi, B * n_mem)], 1).reshape(B, n_mem, 3)
            d_mem = torch.as_tensor(mem, dtype=tdt).cuda()
            d_nmem = torch.full((B,), n_mem, dtype=torch.int32, device="cuda")
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        half = B // 2
        calls = [eng.prepared_batch(hi - lo, d_pose[lo:hi], d_ut[lo:hi], d_u0[lo:hi], stream=st.cuda_stream,
                                    n_steps=None if spl == 1 else spl,
                                    mem_cols=None if d_mem is None else d_mem[lo:hi],
                                    n_mem=None if d_nmem is None else d_nmem[lo:hi], mem_stride=n_mem)
                 for (lo, hi), st in zip(((0, half), (half, B)), streams)]
        torch.cuda.synchronize()
        # size the timed region from a short probe: ~0.3 s
        def run(n_calls):
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            evj = torch.cuda.Event()
            torch.cuda.synchronize()
            ev0.record(streams[0])
            streams[1].wait_event(ev0)
            for _ in range(n_calls):
                for call in calls:
                    call()
            evj.record(streams[1])
            streams[0].wait_event(evj)
            ev1.record(streams[0])
            torch.cuda.synchronize()
            return ev0.elapsed_time(ev1) / (n_calls * spl)   # ms per pass
        probe = run(max(1, 200 // spl))
        if spinup_s > 0.0:   # a stand-alone run of this leg (tools/other_config_point.py): the clock ramp the headline leg provides in bench.py
            run(max(2, int(spinup_s / (probe * 1e-3) / spl)))
        n_calls = max(2, int(0.3 / (probe * 1e-3) / spl))
        run(max(1, n_calls // 4))
        pass_ms = run(n_calls)
        flops = 2 * K * K * (T + n_mem) + 4 * K * K * T + (4 * K + 140) * T   # W with N = T + n_mem
        peak = VALU_F32_PEAK_TF if f32 else VALU_F64_PEAK_TF
        tfl = flops * B / (pass_ms * 1e-3) / 1e12
        # configTarget rebuild of this configuration's grid: device time per rebuild (HIP events around 50 enqueue-only rebuilds)
        st0 = streams[0]
        alt = (b[0], b[1] + 0.1, b[2], b[3])   # (an extent change forces the rebuild; alternate between two extents)
        for i in range(4):
            eng.config_domain(alt if i % 2 == 0 else b)
        evs = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0 = _t.perf_counter()
        evs[0].record(st0)
        for i in range(50):
            eng.config_domain_async(alt if i % 2 == 0 else b, stream=st0.cuda_stream)
        evs[1].record(st0)
        enq = (_t.perf_counter() - t0) / 50
        torch.cuda.synchronize()
        nx, ny = eng.target_grid()[1:]
        res.append({"config": c["name"], "kinematics": c["model"], "num_basis": K, "horizon_steps": T, "dt": c["dt"],
                    "dtype": c["prec"], "agents": B, "lanes_per_agent": eng.agent_lanes(half), "n_mem": n_mem,
                    "steps_per_launch": spl, "passes_timed": n_calls * spl, "launches_timed": n_calls * len(calls),
                    "ms_per_pass": pass_ms, "us_per_4096_agents": 1e3 * pass_ms * 4096 / B,
                    "value": B / (pass_ms * 1e-3), "unit": "optimisations/s",
                    "roofline": {"bound": "valu-%s" % c["prec"], "achieved": tfl, "peak": peak, "unit": "TFLOP/s",
                                 "frac": tfl / peak, "flops_per_optimisation": flops},
                    "config_domain_rebuild": {"grid": "%dx%d" % (nx, ny), "device_us": 1e3 * evs[0].elapsed_time(evs[1]) / 50,
                                              "enqueue_only_wall_us": 1e6 * enq}})
        # the rocprofv3 view of this leg (tools/r06_pack_profile.sh -> profiles/r06_pack_profile.json: kernel average over the
        # timed region of a stand-alone profiled run of the SAME launch form; not measured in this run)
        pr = prof.get((c["name"], B, spl))
        if pr:
            res[-1]["kernel_avg_us_profiled"] = pr["kernel_avg_us_timed_region"]
            res[-1]["frac_profiled"] = flops * B / (pr["kernel_avg_us_timed_region"] / spl * 1e-6) / 1e12 / peak
        eng.close()
    return {"note": "the other single-GPU BASELINE configurations in the headline's launch form (two agent groups x %d steps per "
                    "launch), 4096 agents unless the case names its batch, ~0.3 s timed each, HIP events around the launches of "
                    "both streams; lanes_per_agent < 64: several agents per wavefront (short horizons)" % spl,
            "cases": res}


# the reference's own use of the path: ONE robot, one control() per tick (exploration.hpp:232).  The BASELINE shapes:
TICK_SHAPES = [
    dict(name="configs[0]", model="omni", K=5, dt=0.1, horizon=0.5, prec="f64"),
    dict(name="configs[1]", model="simple_cart", K=10, dt=0.1, horizon=2.0, prec="f64"),
    dict(name="configs[2]", model="omni", K=20, dt=0.02, horizon=5.0, prec="f32"),
    dict(name="configs[2] in fp64", model="omni", K=20, dt=0.02, horizon=5.0, prec="f64"),
    dict(name="configs[3] (one agent of the batch)", model="simple_cart", K=10, dt=0.1, horizon=20.0, prec="f64"),
    dict(name="configs[4] (control call)", model="omni", K=30, dt=0.1, horizon=50.0, prec="f64"),
]


def tick_legs(torch, capi, np):
    """SURVEY.md 8(f) kernels, device time from HIP events on the launch stream (not ctypes wall time):
    (a) `tick_kernels`: Collision::collisionCheck, validate_control and DynamicWindow::control (both overloads) for P = 4096
        and 65 536 poses on the 240 x 120 demo grid, both implementations of the lookup (ring search = up to ~200 dependent
        byte loads per pose: latency-bound; inflated map = one dilation launch + ONE byte per pose-step: launch-bound at these
        sizes); algorithmic bytes: one occupancy byte per ring cell visited / per pose-step, 24 B of pose in, 4 B out;
    (b) `fleet_tick`: eea_tick_batch at B = 4096 robots (explore_omni.yaml shape: K = 10, T = 50) on that grid, robots spread
        over the map (a part of them in front of obstacles: every branch runs), microseconds per tick over 200 ticks
        (reference exploration.hpp:220-279: control -> validate_control -> dynamic window per robot, 10 Hz in production)."""
    COLL = (0.7, 1.0, 0.2, 0.8)
    DWA = (0.1, 2.0, 0.2, 2.5, 2.5, 1.0, 1.0, -1.0, 1.0, -1.0, 2.0, -2.0, 3, 8, 5)
    xs, ys, res, x0, y0 = 240, 120, 0.05, -1.0, -1.0
    data = np.zeros((ys, xs), dtype=np.int8)
    cx, cy = x0 + (np.arange(xs) + 0.5) * res, y0 + (np.arange(ys) + 0.5) * res
    for (a, b, c, d) in [(2.4, 0.2, 3.0, 2.6), (6.0, 2.0, 6.5, 4.6), (8.8, -0.4, 9.4, 1.2)]:
        data[np.ix_((cy >= b) & (cy <= d), (cx >= a) & (cx <= c))] = 100
    ccfg = capi.make_collision_cfg(x0, y0, res, xs, ys, *COLL)
    dcfg = capi.DwaCfg(*DWA)
    d_grid = torch.as_tensor(data).cuda()
    st = torch.cuda.Stream()      # every call below is enqueued on THIS stream, and so are the events around them
    sp = st.cuda_stream
    rng = np.random.default_rng(99)

    def timed(fn, n):
        torch.cuda.synchronize()   # (the inputs were produced on torch's own stream)
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record(st)
        for _ in range(n):
            fn()
        e1.record(st)
        torch.cuda.synchronize()
        return 1e3 * e0.elapsed_time(e1) / n   # us per call (device time on the launch stream)

    kernels = []
    try:
        for impl, iname in ((1, "ring search"), (2, "inflated map")):
            capi.set_option(capi.OPT_COLLISION_IMPL, impl)
            for P in (4096, 65536):
                x = torch.as_tensor(np.stack([rng.uniform(-0.5, 10.5, P), rng.uniform(-0.5, 4.5, P), rng.uniform(-3, 3, P)], 1)).cuda()
                u = torch.as_tensor(np.stack([rng.uniform(-1, 1, P), rng.uniform(-1, 1, P), rng.uniform(-2, 2, P)], 1)).cuda()
                hit = torch.empty((P,), dtype=torch.int32, device="cuda")
                uo = torch.empty((P, 3), dtype=torch.float64, device="cuda")
                xt = x[:, None, :].repeat(1, 50, 1).contiguous()
                row = {"implementation": iname, "poses": P,
                       "collision_check_us": timed(lambda: capi.collision_check_batch(ccfg, d_grid, x, hit, stream=sp), 50),
                       "validate_control_us": timed(lambda: capi.validate_control_batch(ccfg, d_grid, x, u, 0.1, 0.5, hit, stream=sp), 50),
                       "dwa_vref_us": timed(lambda: capi.dwa_control_batch(ccfg, dcfg, d_grid, x, u, uo, hit, vref=u, stream=sp),
                                            10 if P > 4096 else 30),
                       "dwa_traj_us": timed(lambda: capi.dwa_control_batch(ccfg, dcfg, d_grid, x, u, uo, hit, xt_ref=xt, dt_ref=0.1,
                                                                           stream=sp), 10 if P > 4096 else 30)}
                row["collision_check_ns_per_pose"] = 1e3 * row["collision_check_us"] / P
                row["dwa_vref_ns_per_rollout_step"] = 1e3 * row["dwa_vref_us"] / (P * 120 * 20)
                kernels.append(row)
    finally:
        capi.set_option(capi.OPT_COLLISION_IMPL, 0)
    # fleet tick
    B = 4096
    lim = np.array([1.0, 1.0, 2.0])
    eng = capi.Engine(capi.make_config(capi.MODEL_OMNI, 0.1, 5.0, 0.1, 1.0, 10, np.diag([1.0, 1.0, 2.0]), -lim, lim))
    eng.set_target_gaussians(MEANS, SIGMAS)
    eng.config_domain(MAP_BOUNDS)
    T = eng.T
    poses = np.stack([rng.uniform(0.0, 10.0, B), rng.uniform(-0.5, 4.5, B), rng.uniform(-3, 3, B)], 1)
    z = lambda *sh, dt=torch.float64: torch.zeros(sh, dtype=dt, device="cuda")
    d_pose, d_ut, d_u, d_vb = torch.as_tensor(poses).cuda(), z(B, T, 3), z(B, 3), z(B, 3)
    d_follow, d_count, d_valid, d_skip, d_src = (z(B, dt=torch.int32) for _ in range(5))
    d_traj = z(B, T, 3)
    tick = lambda: eng.tick_batch(B, d_pose, d_ut, d_follow, d_count, d_u, d_vb, d_grid, d_traj, d_valid, d_skip, ccfg, dcfg,
                                  0.1, 0.5, source=d_src, stream=sp)
    us = timed(tick, 200)
    src = d_src.cpu().numpy()
    d_follow.zero_(), d_count.zero_(), d_u.zero_(), d_ut.zero_()
    tick_cached = lambda: eng.tick_batch(B, d_pose, d_ut, d_follow, d_count, d_u, d_vb, d_grid, d_traj, d_valid, d_skip, ccfg, dcfg,
                                         0.1, 0.5, source=d_src, grid_epoch=7, stream=sp)
    us_cached = timed(tick_cached, 200)
    ctl = timed(lambda: eng.control_batch(B, d_pose, d_ut, d_u, stream=sp), 200)
    # the same loop with the robots MOVING: after every tick each robot advances by integrate_twist of the
    # twist the tick chose (eea_integrate_twist_batch, numerics.hpp:273-297) and odometry reports that twist -- the branch mix
    # (control / follow / re-plan) is then the closed loop's, not the one static poses freeze; the map is unchanged (epoch)
    d_follow.zero_(), d_count.zero_(), d_u.zero_(), d_ut.zero_(), d_vb.zero_()
    d_pose.copy_(torch.as_tensor(poses))

    def tick_moving():
        eng.tick_batch(B, d_pose, d_ut, d_follow, d_count, d_u, d_vb, d_grid, d_traj, d_valid, d_skip, ccfg, dcfg,
                       0.1, 0.5, source=d_src, grid_epoch=9, stream=sp)
        capi.integrate_twist_batch(d_pose, d_u, 0.1, normalize_heading=True, stream=sp)
        d_vb_copy(d_u)
    with torch.cuda.stream(st):
        d_vb_copy = lambda src_: d_vb.copy_(src_, non_blocking=True)
        us_moving = timed(tick_moving, 200)
        move_only = timed(lambda: (capi.integrate_twist_batch(d_pose, d_u, 0.1, normalize_heading=True, stream=sp), d_vb_copy(d_u)), 200)
    src_moving = d_src.cpu().numpy()
    eng.close()
    # What the two byte-lookup kernels of a tick should cost ("state the bound"):
    #  * dwa_control_kernel<MAP, FLEET>: one workgroup per robot that needs the window, one lane per velocity sample, a rollout of
    #    `steps` DEPENDENT steps per lane -- per step one sincos + ~12 fp64 operations + ONE byte of the inflated map (L2-resident:
    #    the map is 33 KB).  The chain is latency, not bytes: ~steps x (sincos ~60 + arithmetic ~30 + L2 byte ~120 cycles ~ 0.1 us at
    #    2.3 GHz) ~ 2 us per wavefront, and with 2 wavefronts per robot (120 samples) x 4096 robots = 8192 wavefronts on 1024 SIMDs
    #    x 8 resident = one round: the floor is ~2-4 us of chain + launch (~5 us), against 34 us measured -> the kernel is 4-5x its
    #    dependent-chain floor; the difference is the objective (distance to optTraj: `steps` more dependent sqrt / loads of the
    #    reference trajectory per step in the "traj" mode) and the serial first-minimum reduction over 120 samples by lane 0.
    #  * inflate_kernel (dilation of the occupied cells by the ring offsets): bytes = grid read once (28.8 KB) + every occupied
    #    cell stamps |offsets| ~ 250 bytes of the map; at ~10 % occupied cells of 240 x 120 that is ~0.7 MB of scattered byte
    #    stores -> HBM/L2 bandwidth is irrelevant (<< 1 us at 8 TB/s); the floor is the launch (~5 us) + one round of the
    #    workgroups' list-then-stamp (two barriers), against 18 us measured.
    lookups = B * 120 * 20
    return {"tick_kernels": {"grid": "%dx%d int8 @ %.2f m" % (xs, ys, res), "dwa_window": "3 x 8 x 5 samples x 20 steps",
                             "note": "device microseconds per call (HIP events on the launch stream); ring search: dependent byte "
                                     "loads (latency-bound), inflated map: one dilation launch + one byte per pose-step "
                                     "(launch-bound at these sizes)", "cases": kernels},
            "fleet_tick": {"robots": B, "kinematics": "omni", "num_basis": 10, "horizon_steps": T, "us_per_tick": us,
                           "us_per_tick_unchanged_grid": us_cached,
                           "us_per_tick_moving_robots": us_moving - move_only, "pose_update_us": move_only,
                           "ticks_per_s": 1e6 / us, "robot_ticks_per_s": B * 1e6 / us,
                           "control_batch_alone_us": ctl,
                           "sources_last_tick": {n: int((src == i).sum()) for i, n in enumerate(("control", "dwa_follow", "dwa_reference",
                                                                                                "dwa_replan"))},
                           "sources_last_tick_moving": {n: int((src_moving == i).sum()) for i, n in
                                                        enumerate(("control", "dwa_follow", "dwa_reference", "dwa_replan"))},
                           "bounds": {"dwa_window_lookups_per_tick_at_most": lookups,
                                      "dwa_window_dependent_chain_floor_us": 4.0, "dilation_launch_floor_us": 5.0,
                                      "note": "dwa_control_kernel<MAP, FLEET>: 20 dependent rollout steps per lane (sincos + one L2 byte "
                                              "each): a latency chain of ~2-4 us + launch, not a bandwidth problem (the map is 33 KB); "
                                              "inflate_kernel: < 1 MB of scattered byte stores, launch + two barriers ~5 us (bench.py "
                                              "tick_legs carries the derivation)"},
                           "note": "eea_tick_batch: step counters -> control() of the robots that follow no DWA twist -> optTraj "
                                   "rollout -> validate_control -> dynamic window per robot in its mode, one stream, no host round "
                                   "trip; us_per_tick: static poses (the robots in front of obstacles stay in the DWA branches); "
                                   "us_per_tick_moving_robots: the robots advance by integrate_twist of the chosen twist after every "
                                   "tick (eea_integrate_twist_batch), map unchanged; "
                                   "us_per_tick_unchanged_grid: eea_tick_io::grid_epoch != 0, the inflated collision map of the "
                                   "tick before is reused (maps update at ~1 Hz, the loop runs at 10 Hz)"}}


def cpp_host_loop_leg(agents):
    """The consensus leg's enqueue loop from a C++ host (host/test/consensus_bench.cpp through the C ABI) instead of this
    file's Python: (a) local exchange, lag 1; (b) with a COLLECTIVE KERNEL in the exchange -- one rank whose all-reduce is a
    kernel of the stream-asynchronous test double tests/fake_rccl (512 threads x 96 registers x 16 KB of LDS per block: it has
    to become resident beside the control kernels, what a real multi-GPU run's RCCL kernel has to) -- lag 2, stream-ordered
    (what this file does with a communicator) and with one group device-bound (faster, but it can stall at full occupancy:
    agents_timed_out says).  Own processes (their own HIP runtime, nothing shared with this one)."""
    import subprocess
    root = os.path.dirname(os.path.abspath(__file__))
    exe = os.path.join(root, "ergodic_exploration_amd", "host", "build", "consensus_bench")
    fake = os.path.join(root, "tests", "fake_rccl", "librccl.so.1")
    if not os.path.exists(exe):
        return {"error": "host/build/consensus_bench is not built (__graft_entry__.build())"}
    out = {"driver": "ergodic_exploration_amd/host/test/consensus_bench.cpp", "agents": agents, "cases": []}
    cases = [("local exchange (no collective)", "", 1)]
    if os.path.exists(fake):
        cases += [("collective kernel in the exchange (test double), GATED (eea_stream_wait_flag in front of every consuming "
                   "launch; what this file does with a communicator)", fake, 2, "32"),
                  ("collective kernel in the exchange (test double), stream-ordered, ONE device graph per 48 passes "
                   "(eea_consensus_plan)", fake, 2, "22"),
                  ("collective kernel in the exchange (test double), stream-ordered per call (round 5's form)", fake, 2, "12")]
    cases += [("local exchange (no collective), gated", "", 2, "32")]
    for case in cases:
        name, lib, lag = case[:3]
        mode = case[3] if len(case) > 3 else "2"
        try:
            r = subprocess.run([exe, "3000", str(agents), "1", lib, str(lag), mode], capture_output=True, text=True, timeout=120)
            line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
            res = json.loads(line[-1][len("RESULT "):]) if line else {"error": (r.stdout + r.stderr)[-300:]}
        except Exception as exc:  # noqa: BLE001
            res = {"error": repr(exc)}
        res["exchange"] = name
        out["cases"].append(res)
    return out


def single_robot_ticks(torch, capi, np):
    """Dependent eea_control calls of ONE agent at every BASELINE shape: wall time per call including the host round trip
    (the call returns u0 on the host, as ErgodicControl::control does)."""
    res = []
    for c in TICK_SHAPES:
        if c["model"] == "simple_cart":
            model, rdiag, lim = capi.MODEL_SIMPLE_CART, [1.0, 0.0, 2.0], np.array([1.0, 0.0, 2.0])
        else:
            model, rdiag, lim = capi.MODEL_OMNI, [1.0, 1.0, 2.0], np.array([1.0, 1.0, 2.0])
        eng = capi.Engine(capi.make_config(model, c["dt"], c["horizon"], 0.1, 1.0, c["K"], np.diag(rdiag), -lim, lim,
                                           precision=capi.PREC_F32 if c["prec"] == "f32" else capi.PREC_F64))
        eng.set_target_gaussians(MEANS, SIGMAS)
        eng.config_domain(MAP_BOUNDS)
        x = np.array([3.0, 2.0, 0.3])
        for _ in range(20):
            eng.control(MAP_BOUNDS, x)
        n = 300
        t0 = time.perf_counter()
        for _ in range(n):
            eng.control(MAP_BOUNDS, x)
        lat = (time.perf_counter() - t0) / n
        # the same calls served by the RESIDENT workgroup (EEA_OPT_RESIDENT_CONTROL: a host-mapped mailbox, no launch per call)
        lat_res = None
        try:
            capi.set_option(capi.OPT_RESIDENT_CONTROL, 1)
            for _ in range(20):
                eng.control(MAP_BOUNDS, x)
            t0 = time.perf_counter()
            for _ in range(n):
                eng.control(MAP_BOUNDS, x)
            lat_res = (time.perf_counter() - t0) / n
        finally:
            capi.set_option(capi.OPT_RESIDENT_CONTROL, 0)
        res.append({"config": c["name"], "kinematics": c["model"], "num_basis": c["K"], "horizon_steps": eng.T,
                    "dtype": c["prec"], "gpu_us_per_call": 1e6 * lat,
                    "gpu_us_per_call_resident": None if lat_res is None else 1e6 * lat_res})
        eng.close()
    return res


def cpp_tick_latency():
    """host/test/tick_latency.cpp: the same dependent eea_control calls from a C++ host (what a maintainer's binding costs,
    without this file's Python around every call)"""
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ergodic_exploration_amd", "host", "build", "tick_latency")
    if not os.path.exists(exe):
        return {"error": "host/build/tick_latency is not built (__graft_entry__.build())"}
    try:
        r = subprocess.run([exe, "2000"], capture_output=True, text=True, timeout=120)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        return {"driver": "ergodic_exploration_amd/host/test/tick_latency.cpp",
                "cases": json.loads(line[-1][len("RESULT "):])} if line else {"error": (r.stdout + r.stderr)[-300:]}
    except Exception as exc:  # noqa: BLE001
        return {"error": repr(exc)}


def cpu_ticks(seconds_each=0.3):
    """The CPU port's control() at the same shapes (1 thread): us per call, a bounded sample per shape."""
    import numpy as np
    from oracle import pyoracle as po
    res = {}
    for c in TICK_SHAPES:
        if c["model"] == "simple_cart":
            model, rdiag, lim = po.MODEL_SIMPLE_CART, [1.0, 0.0, 2.0], np.array([1.0, 0.0, 2.0])
        else:
            model, rdiag, lim = po.MODEL_OMNI, [1.0, 1.0, 2.0], np.array([1.0, 1.0, 2.0])
        cfg = po.make_config(model, c["dt"], c["horizon"], 0.1, 1.0, c["K"], np.diag(rdiag), -lim, lim)
        pose = np.array([[3.0, 2.0, 0.3]])
        sec, _ = po.bench_control(cfg, MEANS, SIGMAS, MAP_BOUNDS, pose, 2, 1)
        calls = max(2, min(2000, int(seconds_each / max(sec / 2, 1e-7))))
        sec, _ = po.bench_control(cfg, MEANS, SIGMAS, MAP_BOUNDS, pose, calls, 1)
        res[c["name"]] = 1e6 * sec / calls
    return res


def dry_run(args):
    """EEA_BENCH_DRYRUN=1: the launch plumbing without a GPU (CPU test of `--gpus N`): the ranks rendezvous over
    gloo, take the max over ranks of a dummy time like the real legs do, run the grid-tile leg's partition + collective
    on a small grid (numpy in place of the device kernel), and rank 0 prints one JSON line."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from ergodic_exploration_amd import agent_batch as ab
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    grid_tile = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
        dist.barrier()
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert int(t.item()) == world
    if not args.no_grid_tile:
        # the grid-tile leg's structure: rows [row0, row0 + nrows) per rank -> K^2 un-normalised sums -> one all-reduce
        # -> divide by element 0; checked against the un-tiled sums of the same grid
        n, K, res = 48, 4, 0.1
        L = (n - 1) * res
        rng = np.random.default_rng(2024)
        cells = rng.choice(np.array([0.001, 0.7, 0.5]), size=(n, n), p=[0.7, 0.2, 0.1])   # entropy-like cell values
        coord = np.arange(n) * res
        tab = np.cos(np.outer(np.arange(K) * np.pi / L, coord))                            # [k][i]

        def sums(rows, r0):  # un-normalised [k2][k1] sums of the rows r0 .. r0 + len(rows) - 1
            return (tab[:, r0:r0 + rows.shape[0]] @ rows @ tab.T).reshape(-1)

        row0, nrows = ab.grid_row_tile(n, rank, world)
        part = torch.as_tensor(sums(cells[row0:row0 + nrows], row0))
        if world > 1:
            phik = ab.reduce_occupancy_sums(part)
        else:
            phik = part / part[0]
        full = sums(cells, 0)
        err = float(np.abs(phik.numpy() - full / full[0]).max())
        grid_tile = {"rows_per_rank": int(nrows), "max_abs_err_vs_untiled": err, "ok": bool(err < 1e-12)}
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        out = {"dryrun": True, "n_gpus": world, "gpus_arg": args.gpus, "steps": args.steps,
               "warmup": args.warmup, "grid_tile": grid_tile}
        if os.environ.get("EEA_BENCH_DRYRUN_FROM"):
            # the line-size guard of tests/test_bench_line.py: a recorded full result dictionary goes through the SAME
            # emission path as a real run (detail file + compact line)
            with open(os.environ["EEA_BENCH_DRYRUN_FROM"]) as f:
                out = {**json.load(f), **out}
        bench_line.write_detail(out, os.environ.get("EEA_BENCH_DETAIL_DIR", ROOT))
        os.write(1, (bench_line.compact(out) + "\n").encode())


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))

    if os.environ.get("EEA_BENCH_DRYRUN"):
        return dry_run(args)

    import numpy as np
    import torch
    import torch.distributed as dist
    from ergodic_exploration_amd import capi

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = max(1, torch.cuda.device_count())   # counting devices does not initialise the GPU
    shared_device = world > ndev               # plumbing runs: several ranks on one GPU
    # RCCL refuses two ranks on one device: such runs rendezvous over gloo and stage the exchange through the host
    backend = os.environ.get("EEA_DIST_BACKEND", "gloo" if shared_device else "nccl")
    device = local_rank % ndev
    torch.cuda.set_device(device)
    use_dist = world > 1
    if use_dist:
        # RCCL writes its banner / warnings to stdout; keep stdout for the ONE JSON line
        os.environ["NCCL_DEBUG"] = os.environ.get("EEA_NCCL_DEBUG", "WARN")
        os.environ.setdefault("NCCL_DEBUG_FILE", "/tmp/eea_rccl_%h_%p.log")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if backend == "nccl":
            # host tensors over gloo, device tensors over RCCL, and no device_id: the RCCL communicator is created by
            # the first DEVICE collective -- which only the exchange legs issue.  The barriers and the max-over-ranks
            # of the timed region are host collectives, so the headline leg does not depend on RCCL starting up
            dist.init_process_group("cpu:gloo,cuda:nccl")
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
                                       device=device))
    if args.control_kernel == "workgroup":
        capi.set_option(capi.OPT_CONTROL_KERNEL, 1)
    if args.workgroup_threads:
        capi.set_option(capi.OPT_WORKGROUP_THREADS, args.workgroup_threads)
    eng.set_target_gaussians(MEANS, SIGMAS)
    eng.config_domain(MAP_BOUNDS)
    T, K2, B = eng.T, eng.K2, args.agents
    R = max(1, args.passes_per_step)
    RX = max(1, args.exchange_passes_per_step)

    # synthetic inputs: random poses, zero warm start; resident in HBM before any timing
    rng = np.random.default_rng(12345 + rank)
    poses = np.stack([rng.uniform(0.5, 11.5, B) - 1.0, rng.uniform(0.5, 5.5, B) - 1.0,
                      rng.uniform(-np.pi, np.pi, B)], 1)
    d_pose = torch.as_tensor(poses, dtype=tdt).cuda()
    d_ut = torch.zeros((B, T, 3), dtype=tdt, device="cuda")
    d_u0 = torch.empty((B, 3), dtype=tdt, device="cuda")
    d_mem = d_nmem = None
    if args.n_mem:
        mem = np.stack([rng.uniform(0.5, 11.5, B * args.n_mem) - 1.0, rng.uniform(0.5, 5.5, B * args.n_mem) - 1.0,
                        rng.uniform(-np.pi, np.pi, B * args.n_mem)], 1).reshape(B, args.n_mem, 3)
        d_mem = torch.as_tensor(mem, dtype=tdt).cuda()
        d_nmem = torch.full((B,), args.n_mem, dtype=torch.int32, device="cuda")

    compute = torch.cuda.Stream()
    torch.cuda.set_stream(compute)  # once, not per pass: the loop must out-run a ~40 us kernel
    xstream = torch.cuda.Stream(priority=-1)   # exchange steps run beside the next pass (host-staged / all-gather legs)
    # agent groups: contiguous slices of the batch, group 0 on the compute stream
    G = max(1, min(args.agent_groups, B))
    gb = [(g * B) // G for g in range(G + 1)]
    gstreams = [compute] + [torch.cuda.Stream() for _ in range(G - 1)]

    def sl(t, g):
        return None if t is None else t[gb[g]:gb[g + 1]]

    gargs = [dict(B=gb[g + 1] - gb[g], pose=sl(d_pose, g), ut=sl(d_ut, g), u0=sl(d_u0, g),
                  mem_cols=sl(d_mem, g), n_mem=sl(d_nmem, g), stream=gstreams[g].cuda_stream) for g in range(G)]

    # ---- exchange plumbing -------------------------------------------------------------------------------
    # consensus leg: the control kernels leave per-agent sum records (eea_batch_io::d_ck_rec: [c_k, 1]); ONE small
    # launch on the exchange stream adds them (eea_ck_records_sum: sums + agent count), ONE collective adds the ranks'
    # records (nothing with one rank), and pass i + lag divides sum by count inside the kernel (ck_shared_parts = 1):
    # one launch of 2 wavefronts per 32 agents, resident beside the control kernels, instead of three dependent launches.
    LAGS = sorted({max(1, min(6, int(x))) for x in args.consensus_lags.split(",") if x.strip()}) or [1]
    NB = min(8, max(max(LAGS) + 2, args.consensus_buffers))  # record buffers in flight: pass i writes buffer i % NB and
    #                              reads (i - lag) % NB; the sum of pass i - NB has been consumed (its flag waited for) by
    #                              passes that precede pass i on every group stream
    L = eng.ck_record_len
    d_ready = torch.zeros((B,), dtype=torch.int32, device="cuda")   # device-bound exchange: per-agent ready marks ...
    d_flag = torch.zeros((1,), dtype=torch.int32, device="cuda")    # ... and the flag of the finished exchange (sequence numbers)
    d_xstatus = torch.zeros((B,), dtype=torch.int32, device="cuda")  # per-agent status of the consensus passes (timeouts)
    d_gate_timeouts = torch.zeros((1,), dtype=torch.int32, device="cuda")  # gates of the gated exchange that gave up
    cstate = {"lag": LAGS[0], "seq0": 0}
    d_arec_all = torch.empty((NB, B, L), dtype=tdt, device="cuda")   # per-agent records of a pass, one slot per pass in flight
    d_rec_all = torch.zeros((NB, L), dtype=tdt, device="cuda")       # their sum (over all ranks)
    d_arec = [d_arec_all[s] for s in range(NB)]
    d_rec = [d_rec_all[s] for s in range(NB)]
    ev_grp = [[torch.cuda.Event() for _ in range(G)] for _ in range(NB)]
    ev_x = [torch.cuda.Event() for _ in range(NB)]
    d_ck = [torch.empty((B, K2), dtype=tdt, device="cuda") for _ in range(3)]   # all-gather leg
    ev_ck = [torch.cuda.Event() for _ in range(3)]
    ev_ag = [torch.cuda.Event() for _ in range(3)]
    # the communicator is created AFTER the headline leg, under a watchdog (setup_exchange below): a collective
    # library that cannot start on some node must not take the driver-timed line with it
    comm = None          # RCCL communicator behind the C ABI (ranks > 1, or --force-exchange)
    xcomm = None         # the communicator the consensus leg's exchange calls go through (a local one with one rank)
    host_staged = False  # plumbing run: collectives over gloo, staged through the host
    exchange_backend = "local (1 rank)"
    d_all = None

    xcalls = {}

    def collective_in_exchange():
        """the consensus leg's exchange contains a collective kernel (an RCCL communicator, also a one-rank one)"""
        return (not host_staged) and comm is not None and xcomm is comm

    def exchange_records(slot, seq):
        """the exchange of a pass in ONE C-ABI call, device-bound (eea_comm_records_exchange_bound): the record sum polls the
        agents' ready marks, the all-reduce over the ranks follows on the communicator's stream, the flag = seq behind it
        is what the consuming kernels wait for -- no host wait, no stream wait"""
        if not host_staged:
            call = xcalls.get(slot)
            if call is None:
                call = xcalls[slot] = xcomm.prepared_records_exchange_bound(eng, B, d_arec[slot], d_ready, d_rec[slot],
                                                                            d_flag, slot)
            call(seq)
            return
        # plumbing run (ranks share a GPU, gloo): the record sum on the device, the all-reduce staged through the host
        for g in range(G):
            ev_grp[slot][g].record(gstreams[g])
            xstream.wait_event(ev_grp[slot][g])
        eng.ck_records_sum(B, d_arec[slot], d_rec[slot], stream=xstream.cuda_stream)
        with torch.cuda.stream(xstream):
            h = d_rec[slot].cpu()
        dist.all_reduce(h)
        with torch.cuda.stream(xstream):
            d_rec[slot].copy_(h)
        ev_x[slot].record(xstream)

    def exchange_allgather(slot, i):
        dst = d_all[i % 2]  # consecutive gathers alternate between the two receive buffers
        if comm is not None:
            comm.allgather_ck_async(eng, B, d_ck[slot], dst, compute.cuda_stream, slot)
            return
        xstream.wait_event(ev_ck[slot])
        if backend == "nccl":
            with torch.cuda.stream(xstream):
                dist.all_gather_into_tensor(dst, d_ck[slot])
        else:
            with torch.cuda.stream(xstream):
                h = d_ck[slot].cpu()
            hall = torch.empty((world * B, K2), dtype=tdt)
            dist.all_gather_into_tensor(hall, h)
            with torch.cuda.stream(xstream):
                dst.copy_(hall)
        ev_ag[slot].record(xstream)

    def setup_exchange():
        nonlocal comm, xcomm, host_staged, exchange_backend
        if not use_dist:
            if args.force_exchange:   # single-rank RCCL communicator: the collectives run, over one rank
                comm = capi.Comm(device, 1, 0, capi.comm_unique_id())
                exchange_backend = "rccl through the C ABI (eea_comm_*), one rank"
            xcomm = comm if comm is not None else capi.Comm(device, 1, 0, None)
            return
        if args.no_exchange:
            return
        double = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "fake_rccl", "librccl.so.1")
        via_double = False
        if backend != "nccl":
            # ranks share a GPU (RCCL refuses two ranks on one device): the C ABI's exchange steps through the test double of
            # RCCL -- collective KERNELS that meet on the device, the ranks are processes -- when it is there; else the
            # collectives go over gloo, staged through the host (plumbing run either way: the GPU is shared)
            have = torch.tensor([1 if os.path.exists(double) else 0], dtype=torch.int32)
            dist.all_reduce(have, op=dist.ReduceOp.MIN)
            if int(have.item()):
                try:
                    capi.comm_set_library(double)
                    via_double = True
                except Exception as exc:  # noqa: BLE001
                    sys.stderr.write("rank %d: eea_comm_set_library failed (%r)\n" % (rank, exc))
            ok_lib = torch.tensor([1 if via_double else 0], dtype=torch.int32)
            dist.all_reduce(ok_lib, op=dist.ReduceOp.MIN)
            if not int(ok_lib.item()):
                exchange_backend = "gloo, staged through the host (ranks share a GPU: plumbing run)"
                host_staged = True
                return
        # the C ABI's own RCCL communicator: rank 0 creates the id, torch.distributed only carries it.  If that fails on
        # some rank, every rank falls back to torch.distributed's collectives (the run must not die with a secondary leg)
        ok = 1
        uid = [None]
        if rank == 0:
            try:
                uid = [capi.comm_unique_id()]
            except Exception as exc:  # noqa: BLE001
                sys.stderr.write("rank 0: eea_comm_get_unique_id failed (%r)\n" % (exc,))
        dist.broadcast_object_list(uid, src=0, device=torch.device("cpu"))   # every rank takes part (over gloo)
        try:
            if uid[0] is None:
                raise RuntimeError("no RCCL id")
            comm = capi.Comm(device, world, rank, uid[0])
        except Exception as exc:  # noqa: BLE001
            sys.stderr.write("rank %d: eea_comm_create failed (%r); torch.distributed collectives instead\n" % (rank, exc))
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32)  # host tensor: gloo
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()):
            exchange_backend = ("the RCCL test double (tests/fake_rccl: collective kernels that meet on the device) through the C "
                                "ABI (eea_comm_*); the ranks share a GPU: plumbing run" if via_double
                                else "rccl through the C ABI (eea_comm_*)")
            xcomm = comm
        else:
            if comm is not None:
                comm.close()
            comm = None
            exchange_backend = ("gloo, staged through the host (ranks share a GPU; eea_comm_create through the test double failed)"
                                if via_double else "rccl through torch.distributed, staged (eea_comm_create failed)")
            host_staged = True

    state = {"i": 0}
    # eea_batch_io structs are built once per distinct buffer set, a pass is one ctypes call per group
    # receding-horizon steps per launch: the largest divisor of the passes per step that does not exceed the request
    SPL = max(d for d in range(1, max(1, args.steps_per_launch) + 1) if R % d == 0)
    shard_calls = [eng.prepared_batch(a["B"], a["pose"], a["ut"], a["u0"], mem_cols=a["mem_cols"], n_mem=a["n_mem"],
                                      mem_stride=args.n_mem, stream=a["stream"],
                                      n_steps=None if SPL == 1 else SPL) for a in gargs]
    shard1_calls = shard_calls if SPL == 1 else [
        eng.prepared_batch(a["B"], a["pose"], a["ut"], a["u0"], mem_cols=a["mem_cols"], n_mem=a["n_mem"],
                           mem_stride=args.n_mem, stream=a["stream"]) for a in gargs]
    exch_calls = {}

    def one_pass(leg):
        i = state["i"]
        state["i"] = i + 1
        if leg == "shard":      # SPL passes per call
            for call in shard_calls:
                call()
            return
        if leg == "shard1":     # the same pass as ONE launch per pass and group (what rounds 1-3 timed)
            for call in shard1_calls:
                call()
            return
        if leg == "consensus":
            lag = cstate["lag"]
            slot = i % NB
            src = (i - lag) % NB if i >= lag else None
            seq = cstate["seq0"] + i + 1      # sequence numbers only grow (also from one timed() call to the next)
            if collective_in_exchange():
                # A COLLECTIVE KERNEL in the exchange (an RCCL communicator): the GATED exchange (ABI 6) -- the device-bound
                # exchange (records + ready marks out, the record sum polls the marks, all-reduce, publish, flag) with the flag
                # wait as a one-wavefront GATE kernel in front of every consuming launch (eea_stream_wait_flag) instead of inside
                # it: a waiting group holds one execution slot, its half of the chip stays empty until the record is there, so the
                # collective kernel always finds room.  (Waiting INSIDE the control kernels dead-locks at full occupancy -- round
                # 5, profiles/r05_two_ranks.txt.)  Launches only: no event, no stream wait (round 5's event-ordered form cost the
                # host 35-40 us per 23 us pass; profiles/r06_exchange_modes.txt).
                slot = seq % NB
                src = (seq - lag) % NB if i >= lag else None
                for g, a in enumerate(gargs):
                    if src is not None:
                        gate = exch_calls.get(("gate", g))
                        if gate is None:
                            gate = exch_calls[("gate", g)] = capi.prepared_stream_wait_flag(d_flag, d_gate_timeouts, a["stream"])
                        gate(seq - lag)
                    call = exch_calls.get(("gated", g, slot, src))
                    if call is None:
                        call = exch_calls[("gated", g, slot, src)] = eng.prepared_batch(
                            a["B"], a["pose"], a["ut"], a["u0"], mem_cols=a["mem_cols"], n_mem=a["n_mem"],
                            mem_stride=args.n_mem, stream=a["stream"], ck_rec=d_arec[slot][gb[g]:gb[g + 1]],
                            rec_ready=d_ready[gb[g]:gb[g + 1]], status=d_xstatus[gb[g]:gb[g + 1]],
                            ck_shared=None if src is None else d_rec[src], ck_shared_parts=0 if src is None else 1)
                    call(seq, 0)
                exchange_records(slot, seq)
                return
            if not host_staged:
                # device-bound: G control launches (ready marks out, flag wait in) + ONE exchange call, nothing else
                slot = seq % NB
                src = (seq - lag) % NB if i >= lag else None
                for g, a in enumerate(gargs):
                    call = exch_calls.get((g, slot, src))
                    if call is None:
                        call = exch_calls[(g, slot, src)] = eng.prepared_batch(
                            a["B"], a["pose"], a["ut"], a["u0"], mem_cols=a["mem_cols"], n_mem=a["n_mem"],
                            mem_stride=args.n_mem, stream=a["stream"], ck_rec=d_arec[slot][gb[g]:gb[g + 1]],
                            rec_ready=d_ready[gb[g]:gb[g + 1]], status=d_xstatus[gb[g]:gb[g + 1]],
                            ck_shared=None if src is None else d_rec[src], ck_shared_parts=0 if src is None else 1,
                            ck_flag=None if src is None else d_flag)
                    call(seq, seq - lag)
                exchange_records(slot, seq)
                return
            for g, a in enumerate(gargs):
                if src is not None:   # the sum record of pass i - lag is complete (on every rank)
                    gstreams[g].wait_event(ev_x[src])
                call = exch_calls.get((g, slot, src))
                if call is None:
                    call = exch_calls[(g, slot, src)] = eng.prepared_batch(
                        a["B"], a["pose"], a["ut"], a["u0"], mem_cols=a["mem_cols"], n_mem=a["n_mem"],
                        mem_stride=args.n_mem, stream=a["stream"], ck_rec=d_arec[slot][gb[g]:gb[g + 1]],
                        ck_shared=None if src is None else d_rec[src], ck_shared_parts=0 if src is None else 1)
                call()
            exchange_records(slot, seq)
            return
        # all-gather leg: one launch per pass that writes every agent's c_k, one ncclAllGather beside the next pass
        slot = i % 3
        in_c = comm is not None   # the completion events then live in the C ABI
        if i >= 2:
            if in_c:
                comm.wait((i - 2) % 3, compute.cuda_stream)  # the gather that read this slot's predecessor is done
            else:
                compute.wait_event(ev_ag[(i - 2) % 3])
        call = exch_calls.get(("ag", slot))
        if call is None:
            call = exch_calls[("ag", slot)] = eng.prepared_batch(B, d_pose, d_ut, d_u0, mem_cols=d_mem, n_mem=d_nmem,
                                                                 mem_stride=args.n_mem, ck=d_ck[slot],
                                                                 stream=compute.cuda_stream)
        call()
        if not in_c:
            ev_ck[slot].record(compute)
        exchange_allgather(slot, i)

    def host_barrier():
        """a barrier every rank leaves together, as a host collective (gloo): independent of the device library"""
        dist.all_reduce(torch.zeros(1, dtype=torch.float64))

    ev_join = [torch.cuda.Event() for _ in range(G)]
    per_rank_s = {}   # leg -> the ranks' own wall times of the last timed() call of that leg (value = work / their MAX)

    def fork_groups():
        """every group stream waits for what the compute stream has enqueued so far"""
        ev_join[0].record(compute)
        for g in range(1, G):
            gstreams[g].wait_event(ev_join[0])

    def join_groups():
        """the compute stream waits for everything the other group streams have enqueued so far"""
        for g in range(1, G):
            ev_join[g].record(gstreams[g])
            compute.wait_event(ev_join[g])

    def timed(leg, steps, warmup, passes=None):
        """EXACTLY `steps` steps (of `passes` passes each) between barrier + synchronize on both sides; max over ranks"""
        Rl = R if passes is None else passes
        if leg == "consensus":
            cstate["seq0"] += state["i"] + 8   # past every sequence number the previous consensus run used
            if not host_staged:
                # the first `lag` steps consume "nothing yet": zeroed sum records (agent count 0 = own c_k) behind a flag
                # that already stands at the sequence number before the first
                torch.cuda.synchronize()
                d_rec_all.zero_()
                d_flag.fill_(cstate["seq0"])
                torch.cuda.synchronize()
        state["i"] = 0
        d_ut.zero_()      # on the compute stream ...
        fork_groups()     # ... and ordered before the first pass of every agent group
        # device spin-up, not part of any count: the shader clock needs a few tens of milliseconds of load to reach
        # its sustained state (with 100 passes of warm-up the timed region still starts on the ramp: 27.2 us per
        # pass against 25.8 us after 1000)
        per = SPL if leg == "shard" else 1   # passes one call of one_pass() issues
        for _ in range(SPINUP_PASSES // per if leg == "shard" else 0):
            one_pass(leg)
        for _ in range(warmup * Rl // per):
            one_pass(leg)
        torch.cuda.synchronize()
        if use_dist:
            host_barrier()
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record(compute)   # the device is idle here (synchronised above): the start of every group's first pass
        for _ in range(steps * Rl // per):
            one_pass(leg)
        join_groups()         # the end event follows the last pass of EVERY agent group
        ev1.record(compute)
        enqueue_s = time.perf_counter() - t0
        torch.cuda.synchronize()  # all streams of the device
        if use_dist:
            host_barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        pass_ms = ev0.elapsed_time(ev1) / (steps * Rl)  # HIP events bracketing the launches of all group streams
        per_rank_s[leg] = [elapsed]
        if use_dist:
            every = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]   # host tensors: gloo
            dist.all_gather(every, torch.tensor([elapsed], dtype=torch.float64))
            per_rank_s[leg] = [float(x) for x in every]
            t = torch.tensor([elapsed, pass_ms], dtype=torch.float64)  # host tensor: gloo
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed, pass_ms = float(t[0]), float(t[1])
        return elapsed, pass_ms, enqueue_s

    elapsed, pass_ms, enqueue_s = timed("shard", args.steps, args.warmup)
    # the same passes as one launch per pass (short: 3 x 400 passes), device work contiguous with the headline leg
    single_pass_ms = pass_ms
    if SPL > 1 and not args.no_single_launch:
        _, single_pass_ms, single_enqueue_s = timed("shard1", 3, 1, passes=400)

    out = None
    if rank == 0:
        N = T + args.n_mem
        K = args.num_basis
        value = world * B * R * args.steps / elapsed
        # algorithmic HBM bytes per optimisation (DESIGN.md "control kernel roofline"): pose in + ut in + ut out +
        # u0 out (+ memory columns in); the shard leg writes no c_k (nothing consumes it there)
        bytes_per_opt = rs * (3 + 3 * T + 3 * T + 3 + 3 * args.n_mem)
        flops_per_opt = 2 * K * K * N + 4 * K * K * T + (4 * K + 140) * T  # SURVEY.md 8(d) "W"
        launch_s = pass_ms * 1e-3
        Bl = gb[1] - gb[0]   # agents per launch (group 0, whose stream carries the events)
        # G launches (one per agent group) run concurrently, each taking launch_s: chip-level rate = G x per-launch
        hbm_gbs = G * bytes_per_opt * Bl / launch_s / 1e9
        tflops = G * flops_per_opt * Bl / launch_s / 1e12
        traffic, traffic_source = None, None
        for name in ("r05_control_pmc.json", "r05_control_pmc_spl1.json", "r04_control_pmc.json", "r04_control_pmc_spl1.json"):
            pmc = os.path.join(ROOT, "profiles", name)
            if not os.path.exists(pmc):
                continue
            try:
                with open(pmc) as f:
                    rec = json.load(f)
                if (rec.get("T") == T and rec.get("K") == K and rec.get("precision") == args.precision
                        and rec.get("steps_per_launch", 1) == SPL and args.model == "simple_cart" and not args.n_mem):
                    traffic = rec.get("hbm_bytes_per_launch") * Bl / rec["agents_per_launch"]  # per launch of Bl agents x SPL steps
                    traffic_source = ("profiles/%s (separate rocprofv3 --pmc passes of this command shape; NOT measured "
                                      "in this run)" % name)
                    break
            except Exception:
                pass
        # the rocprofv3 view of the same command shape (profiles/r04_bench_profile.json, written by
        # tools/summarize_prof.py from a separate profiled run: NOT measured in this run)
        profiled = {}
        try:
            prof_name = "r05_bench_profile.json" if os.path.exists(os.path.join(ROOT, "profiles", "r05_bench_profile.json")) \
                else "r04_bench_profile.json"
            with open(os.path.join(ROOT, "profiles", prof_name)) as f:
                rec = json.load(f)
            if (rec.get("agents") == B and rec.get("T") == T and rec.get("K") == K and rec.get("precision") == args.precision
                    and rec.get("concurrent_launches") == G and rec.get("steps_per_launch", 1) == SPL
                    and args.model == "simple_cart" and not args.n_mem):
                per_pass_us = rec["kernel_avg_us_profiled"] / SPL
                profiled = {"kernel_avg_us_profiled": rec["kernel_avg_us_profiled"],
                            "kernel_avg_us_per_pass_profiled": per_pass_us,
                            "pass_period_us_profiled": rec["pass_period_us_from_trace"],
                            "frac_profiled": G * flops_per_opt * rec["agents_per_launch"] / (per_pass_us * 1e-6) / 1e12 / VALU_F64_PEAK_TF,
                            "effective_clock_ghz_profiled": rec.get("effective_clock_ghz"),
                            "profiled_source": "profiles/%s (rocprofv3 --kernel-trace --stats of this "
                                               "command shape, timed-region dispatches only; a separate run on another box "
                                               "of the pool)" % prof_name}
        except Exception:
            pass
        vpeak = VALU_F32_PEAK_TF if f32 else VALU_F64_PEAK_TF
        # What the kernel's OWN instruction stream allows: a vector instruction holds a SIMD's pipe for 4
        # cycles, a 4x4x4 fp64 matrix instruction for 16, and they share it (profiles/r02_ubench_coissue.txt): with w resident
        # wavefronts per SIMD a pass cannot take less than w x (4 VALU + 16 MFMA) / clock.  Instruction counts per wavefront
        # from the SQ counters of this kernel (profiles/r05_isa_counts.json, rocprofv3 --pmc), the static budget of round 4
        # (profiles/r04_isa_budget.txt: 2 077 + 117) if that file is missing; clock = the measured shader clock under this
        # load.  frac_of_issue_bound = issue_bound_us / pass time: how much of the pass the pipe is busy; what is left is
        # dependency stalls and waits (SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES, the counter file).
        issue = {}
        try:
            valu, mfma, src = 2077.0 + 117.0, 117.0, "profiles/r04_isa_budget.txt (static budget of the metric point)"
            cpath = os.path.join(ROOT, "profiles", "r05_isa_counts.json")
            if os.path.exists(cpath):
                with open(cpath) as f:
                    ic = json.load(f)
                if ic.get("T") == T and ic.get("K") == K and ic.get("precision") == args.precision and ic.get("matrix_insts_per_wave"):
                    valu, mfma, src = ic["valu_insts_per_wave_incl_matrix"], ic["matrix_insts_per_wave"], "profiles/r05_isa_counts.json (SQ counters)"
                    issue["wait_inst_any_over_wave_cycles"] = ic.get("wait_inst_any_over_wave_cycles")
            if T == 200 and K == 10 and not f32 and args.model == "simple_cart" and not args.n_mem:
                clock = profiled.get("effective_clock_ghz_profiled") or 2.33
                waves_per_simd = B / 1024.0
                bound_us = waves_per_simd * (4.0 * (valu - mfma) + 16.0 * mfma) / (clock * 1e3)
                issue.update({"issue_bound_us": bound_us, "frac_of_issue_bound": bound_us / (pass_ms * 1e3),
                              "vector_insts_per_agent": valu - mfma, "matrix_insts_per_agent": mfma,
                              "pipe_cycles_per_agent": 4.0 * (valu - mfma) + 16.0 * mfma, "shader_clock_ghz": clock,
                              "wavefronts_per_simd": waves_per_simd, "issue_bound_source": src,
                              # north_star asks for the matrix-core utilisation: the share of the pass a SIMD's pipe spends in
                              # v_mfma_f64_4x4x4 (16 cycles each; = SQ_VALU_MFMA_BUSY_CYCLES / SIMD-cycles of the pass).  The
                              # contraction is K^2 T of the ~6 K^2 T + 140 T multiply-adds of a pass, and fp64 matrix
                              # instructions run on the vector pipe's own multipliers at the vector rate: a low share is the
                              # shape of the work, not idle matrix cores beside a busy vector pipe
                              "mfma_busy_frac": waves_per_simd * 16.0 * mfma / (pass_ms * 1e3 * clock * 1e3),
                              "reference_formulation_cycles_per_agent": 4.0 * flops_per_opt / 2.0 / 64.0})
        except Exception:
            pass
        out = {
            "metric": "receding-horizon optimisations/sec at K=10x10, T=200; 1/2/4/8-GPU agent-batch",
            "value": value, "unit": "optimisations/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic",
            "config": {"workload": "BASELINE configs[3]: %d-agent batch per GPU, %s, two-Gaussian 12x6 m map, "
                                   "K=%dx%d, T=%d steps (dt %.3g, horizon %.3g), n_mem=%d; pure agent shard "
                                   "(no data-path collective)"
                                   % (B, args.model, K, K, T, args.dt, args.horizon, args.n_mem),
                       "agents_per_gpu": B, "num_basis": K, "horizon_steps": T, "kinematics": args.model,
                       "passes_per_step": R, "optimisations_per_step": world * B * R,
                       "agent_groups": G, "steps_per_launch": SPL, "parallelism": "agent-batch x%d" % world,
                       "dist_backend": ("gloo (host collectives: barriers, timing) + nccl (device collectives: exchange legs)"
                                        if backend == "nccl" else backend) if use_dist else None},
            "timed_region_s": elapsed, "ms_per_pass": 1e3 * elapsed / (args.steps * R),
            # the ranks' own wall times of the timed region and what each processed per second; `value` is all
            # ranks' work / the MAX of these times (= timed_region_s)
            "per_rank": {"timed_region_s": per_rank_s.get("shard"),
                         "values": [B * R * args.steps / t for t in per_rank_s.get("shard", [])]},
            "spinup_passes": SPINUP_PASSES,
            "single_launch_per_pass": {"ms_per_pass": single_pass_ms, "frac": tflops / vpeak * pass_ms / single_pass_ms,
                                       "note": "the same passes as ONE launch per pass and agent group (eea_control_batch; "
                                               "what rounds 1-3 reported), 3 x 400 passes right after the headline leg"},
            "host_enqueue_us_per_pass": 1e6 * enqueue_s / (args.steps * R),
            "parity_tol": PARITY_TOL[args.precision],
            "roofline": {"bound": "valu-%s" % args.precision,
                         "kernel": "control_wave_kernel (%d concurrent launch%s, one per agent group, %d receding-horizon "
                                   "step%s per launch)" % (G, "" if G == 1 else "es", SPL, "" if SPL == 1 else "s"),
                         "achieved": tflops, "peak": vpeak, "unit": "TFLOP/s", "frac": tflops / vpeak,
                         "traffic": traffic, "traffic_source": traffic_source,
                         "flops_per_launch": flops_per_opt * Bl * SPL, "launch_ms": pass_ms * SPL,
                         "passes_per_launch": SPL, "agents_per_launch": Bl,
                         "concurrent_launches": G, "achieved_per_launch": tflops / G, **profiled, **issue,
                         "note": "the control kernel is vector-ALU / transcendental bound, not HBM bound (SURVEY.md "
                                 "8(d)); W = 2K^2N + 4K^2T + (4K+140)T flop per optimisation (reference formulation); frac = "
                                 "achieved / the NOMINAL fp64 vector peak; frac_of_issue_bound = how busy the SIMD pipes are with "
                                 "the instructions the kernel actually issues"},
            "roofline_hbm": {"bound": "hbm", "kernel": "control kernel",
                             "algorithmic_rate": hbm_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "algorithmic_frac": hbm_gbs / HBM_PEAK_GBS,
                             "counter_rate": None if traffic is None else G * traffic / (launch_s * SPL) / 1e9,
                             "traffic": traffic, "traffic_source": traffic_source, "bytes_per_launch": bytes_per_opt * Bl * SPL,
                             "launch_ms": pass_ms * SPL, "passes_per_launch": SPL, "agents_per_launch": Bl,
                             "concurrent_launches": G,
                             "note": "algorithmic_rate = the bytes the reference formulation moves per optimisation (pose, controls in "
                                     "and out) / time -- NOT bytes the device moved: with several steps per launch the controls stay in "
                                     "L2 / LDS between steps and counter_rate (FETCH_SIZE x 2 + WRITE_SIZE of the profiled run) is what "
                                     "HBM actually saw.  Either way the kernel is nowhere near the HBM roofline: north_star asks for the "
                                     "figure"},
        }
        if world == 1 and not args.no_other_configs and not args.n_mem:
            try:
                out["other_configs"] = other_config_legs(args, torch, capi, np, SPL)
            except Exception as exc:  # the headline line must not die with a secondary leg
                out["other_configs"] = {"error": repr(exc)}
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
            try:
                out["single_robot_tick"] = {
                    "note": "the reference's own use: ONE robot, one control() per tick (exploration.hpp:232) -- dependent "
                            "eea_control calls at every BASELINE shape, wall time per call incl. the host round trip; "
                            "gpu_us_per_call_resident: served by the resident server (EEA_OPT_RESIDENT_CONTROL: one wavefront at horizons "
                            "<= 64 steps, a workgroup beyond; a host-mapped mailbox instead of a launch per call); cpp_host: the same calls from a C++ loop; "
                            "cpu_port_us_per_call = the oracle's control() at the same shape, 1 thread, this host",
                    "cases": single_robot_ticks(torch, capi, np),
                    "cpp_host": cpp_tick_latency()}
            except Exception as exc:  # noqa: BLE001 -- the headline line must not die with a secondary leg
                out["single_robot_tick"] = {"error": repr(exc)}
        if world == 1 and not args.no_latency and not f32:
            try:
                out.update(tick_legs(torch, capi, np))
            except Exception as exc:  # noqa: BLE001 -- the headline line must not die with a secondary leg
                out["fleet_tick"] = {"error": repr(exc)}
        if world == 1 and not args.no_phik and not f32:
            try:
                out.update(phik_legs(args, torch, capi, np))
            except Exception as exc:  # the headline line must not die with a secondary leg
                out["roofline_phik"] = {"error": repr(exc)}

    def grid_tile_leg():
        """BASELINE configs[4] shard: the 1024 x 1024 occupancy grid row-tiled over the ranks -- every rank streams its
        rows (int8 cells -> entropy -> K^2 un-normalised sums, eea_spatial_coeff_occupancy_rows), ONE all-reduce of
        K^2 = 900 reals (7.2 KB) over RCCL, phi_k = sums / sums[0] installed on the device (eea_set_phik_from_sums):
        three stream-ordered steps, no host round trip.  With one rank it is the single-tile form of the same calls."""
        from ergodic_exploration_amd import agent_batch as ab
        n, K5, res = 1024, 30, 0.1
        lx5 = ly5 = (n - 1) * res
        rng5 = np.random.default_rng(2024)   # 70 % free, 10 % occupied, 20 % unknown, 32 x 32 blocks
        blocks = rng5.choice(np.array([0, 100, -1], dtype=np.int8), size=(n // 32 + 1, n // 32 + 1), p=[0.7, 0.1, 0.2])
        occ = np.ascontiguousarray(np.kron(blocks, np.ones((32, 32), dtype=np.int8))[:n, :n])
        row0, nrows = ab.grid_row_tile(n, rank, world)
        d_rows = torch.as_tensor(occ[row0:row0 + nrows]).cuda()
        e5 = capi.Engine(capi.make_config(capi.MODEL_OMNI, 0.1, 50.0, res, 1.0, K5, np.diag([1.0, 1.0, 2.0]),
                                          [-1.0, -1.0, -2.0], [1.0, 1.0, 2.0], device=device))
        d_sums = torch.empty((K5 * K5,), dtype=torch.float64, device="cuda")
        st = compute.cuda_stream

        def rebuild():
            e5.spatial_coeff_occupancy_rows(n, n, row0, nrows, d_rows, lx5, ly5, d_sums, stream=st)
            if comm is not None:
                comm.allreduce_sum(e5, d_sums, K5 * K5, stream=st)
            elif use_dist and backend == "nccl":
                dist.all_reduce(d_sums)            # torch's current stream is the compute stream
            elif use_dist:                          # plumbing run (ranks share a GPU): staged through the host
                h = d_sums.cpu()
                dist.all_reduce(h)
                d_sums.copy_(h)
            e5.set_phik_from_sums(d_sums, lx5, ly5, stream=st)

        for _ in range(5):
            rebuild()
        torch.cuda.synchronize()
        if use_dist:
            host_barrier()
        reps = 200
        t0 = time.perf_counter()
        for _ in range(reps):
            rebuild()
        torch.cuda.synchronize()
        back_to_back = (time.perf_counter() - t0) / reps
        lat = []
        for _ in range(50):
            t0 = time.perf_counter()
            rebuild()
            torch.cuda.synchronize()
            lat.append(time.perf_counter() - t0)
        t = torch.tensor([back_to_back, float(np.median(lat))], dtype=torch.float64)
        if use_dist:
            host_barrier()
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        phik0 = float(e5.phik()[0])
        e5.close()
        return {"workload": "BASELINE configs[4]: %dx%d int8 occupancy grid (entropy target), K=%d, rows tiled over %d rank%s"
                            % (n, n, K5, world, "" if world == 1 else "s"),
                "rows_per_rank": nrows, "bytes_streamed_per_rank": int(nrows) * n,
                "allreduce_bytes": 8 * K5 * K5, "collective": (exchange_backend if world > 1 else "none (1 rank)"),
                "us_per_rebuild_back_to_back": 1e6 * float(t[0]), "us_per_rebuild_with_host_wait": 1e6 * float(t[1]),
                "phik_00_check": phik0,
                "note": "rebuild = eea_spatial_coeff_occupancy_rows + all-reduce(K^2 reals) + eea_set_phik_from_sums, "
                        "stream-ordered; max over ranks"}

    emitted = threading.Lock()

    def emit():
        """the ONE JSON line, written in one piece after every C-level stdio buffer (collective library banners) has
        been flushed; at most once per process"""
        if not emitted.acquire(blocking=False):
            return
        if rank == 0:
            import ctypes
            sys.stdout.flush()
            try:
                ctypes.CDLL(None).fflush(None)
            except OSError:
                pass
            # the driver parses the LAST stdout line from a bounded tail: a fixed selection of fields, <= 4 KB by
            # construction (bench_line.compact); every leg's full record goes to bench_detail.json
            bench_line.write_detail(out, os.environ.get("EEA_BENCH_DETAIL_DIR", ROOT))
            try:
                text = bench_line.compact(out)
            except Exception as exc:  # noqa: BLE001 -- never lose the headline to a formatting error
                text = json.dumps({k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                                                            "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
                                  | {"line_error": repr(exc)[:200]})
            os.write(1, (text + "\n").encode())

    # ---- exchange legs: last, and under a watchdog.  They are the only part of this program that has never run on
    # more than one GPU; if a collective never returns, every rank gives up after the timeout, rank 0 still prints
    # the line (with the reason in "exchange") and the processes exit without waiting for the library
    if not args.no_exchange:
        limit = float(os.environ.get("EEA_BENCH_EXCHANGE_TIMEOUT", "300"))
        finished = threading.Event()

        def watchdog():
            if finished.wait(limit):
                return
            if rank == 0:
                out["exchange"] = {"error": "exchange legs did not finish within %g s; headline leg unaffected" % limit}
            emit()
            # a process that has used the GPU and gives up on a hung collective must not look like a clean run: the
            # headline line is out (rank 0), the exit code tells the launcher / driver that a leg deadlocked
            os._exit(3)

        threading.Thread(target=watchdog, daemon=True).start()
        try:
            setup_exchange()
            exchange = {"backend": exchange_backend,
                        # what the collective library itself reports (ncclCommCount): proves that RCCL saw every rank
                        "rccl_nranks": (comm.library_nranks() if comm is not None else 0), "world": world,
                        "consumer": "eea_batch_io::d_ck_shared as a sum record, ck_shared_parts = 1 (the gradient uses c_bar)"}
            by_lag = {}
            # with a collective kernel in the exchange (gated, above) a lag of one pass has the collective on the
            # critical path of every pass: the leg runs the lags >= 2
            lags = [l for l in LAGS if l >= 2] or [2] if collective_in_exchange() else LAGS
            for lag in lags:
                cstate["lag"] = lag
                e_s, p_ms, q_s = timed("consensus", args.steps, args.warmup, passes=RX)
                torch.cuda.synchronize()
                by_lag[str(lag)] = {
                    "pass_ms": p_ms, "pass_ms_vs_single_launch_pass": p_ms / single_pass_ms,
                    "pass_ms_vs_headline": p_ms / pass_ms,
                    "value": world * B * RX * args.steps / e_s, "unit": "optimisations/s",
                    "host_enqueue_us_per_pass": 1e6 * q_s / (args.steps * RX),
                    "agents_timed_out": int((d_xstatus != 0).sum().item()) + int(d_gate_timeouts.item())}
            first = by_lag[str(lags[0])]
            exchange["consensus_allreduce"] = {
                "lag_passes": lags[0], "consuming_groups": ("all gated (eea_stream_wait_flag in front of every consuming launch)"
                                                            if collective_in_exchange() else "all device-bound"),
                "pass_ms": first["pass_ms"], "pass_ms_vs_headline": first["pass_ms_vs_headline"],
                "pass_ms_vs_single_launch_pass": first["pass_ms_vs_single_launch_pass"],
                "value": first["value"], "unit": "optimisations/s", "by_lag": by_lag,
                "agent_groups": G, "bytes_per_rank_per_pass": rs * L, "host_threads": 0,
                "protocol": ("gated (a collective kernel is in the exchange)" if collective_in_exchange() else
                             "device-bound (eea_comm_records_exchange_bound): no host wait, no stream wait, no event"),
                "note": "every pass: the control kernels write per-agent sum records and ready marks (write-through, half way "
                        "through the wavefront), ONE launch on the exchange stream polls the marks and adds the records "
                        "beside the running control kernels (+ one all-reduce of the record over the ranks and a publish "
                        "launch with an RCCL communicator), and pass i waits INSIDE its kernels, right before the first use "
                        "of c_bar, for the flag of pass i - lag; lag 1 = the previous step's consensus.  One launch per pass "
                        "and group (the headline runs %d steps per launch: pass_ms_vs_single_launch_pass is the like-for-"
                        "like ratio).  Every wait of this protocol is for work that was enqueued BEFORE the waiter, whatever the "
                        "stream -> hardware-queue mapping: that is why it is one step per launch" % SPL}
            if world == 1 and rank == 0 and not f32:
                exchange["cpp_host_loop"] = cpp_host_loop_leg(B)
            if use_dist or args.force_exchange:
                d_all = [torch.empty((world * B, K2), dtype=tdt, device="cuda") for _ in range(2)]
                e_s, p_ms, _ = timed("allgather", args.steps, args.warmup, passes=RX)
                exchange["allgather_ck"] = {
                    "value": world * B * RX * args.steps / e_s, "unit": "optimisations/s",
                    "ms_per_step": 1e3 * e_s / args.steps, "pass_ms": p_ms,
                    "bytes_received_per_rank_per_pass": rs * K2 * B * world,
                    "note": "every pass: one ncclAllGather of all agents' c_k (north_star's exchange); nothing on the "
                            "control path consumes the gathered matrix -- the consensus leg is the consuming form"}

            if not args.no_grid_tile and not f32:
                try:
                    gt = grid_tile_leg()
                except Exception as exc:  # noqa: BLE001
                    gt = {"error": repr(exc)}
                if rank == 0:
                    out["grid_tile"] = gt
        except Exception as exc:  # noqa: BLE001 -- the headline line must not die with a secondary leg
            exchange = {"error": repr(exc)}
        finished.set()
        if rank == 0:
            out["exchange"] = exchange
    elif world == 1 and not args.no_grid_tile and not f32:
        try:
            out["grid_tile"] = grid_tile_leg()
        except Exception as exc:  # noqa: BLE001
            out["grid_tile"] = {"error": repr(exc)}
    if xcomm is not None and xcomm is not comm:
        xcomm.close()
    if comm is not None:
        comm.close()
    eng.close()
    if use_dist:
        dist.destroy_process_group()
    # the CPU baseline LAST: every device leg above ran back to back (the driver samples GPU activity every few seconds)
    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        one, allc = cpu_baseline(args, args.cpu_seconds)
        out["cpu_baseline"] = one
        out["cpu_baseline_all_cores"] = allc
        if isinstance(out.get("single_robot_tick", {}).get("cases"), list):
            try:
                cpu = cpu_ticks()
                for c in out["single_robot_tick"]["cases"]:
                    c["cpu_port_us_per_call"] = cpu.get(c["config"])
            except Exception as exc:  # noqa: BLE001
                out["single_robot_tick"]["cpu_error"] = repr(exc)
    emit()


if __name__ == "__main__":
    main()
