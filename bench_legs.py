"""The legs of bench.py that stand on their own (no closure over the headline leg's state): host description, the CPU baseline
(oracle: the ONLY use of oracle/ outside tests/ and smoke()), the phi_k legs, the other BASELINE shapes, the tick kernels and
the fleet tick, the C++ host loops, the single-robot ticks.  bench.py imports them; tools/other_config_point.py runs one of the
other-config legs as a stand-alone program (rocprofv3).  Split out of bench.py in round 6."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MAP_BOUNDS = (-1.0, 11.0, -1.0, 5.0)
MEANS = [[2.5, 2.5], [8.5, 2.5]]
SIGMAS = [[1.5, 1.5], [1.5, 1.5]]
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s
VALU_F64_PEAK_TF = 78.6     # fp64 vector peak
VALU_F32_PEAK_TF = 157.3


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
        pname = next((n for n in ("r06_phik_pmc.json", "r05_phik_pmc.json", "r03_phik_pmc.json") if os.path.exists(os.path.join(ROOT, "profiles", n))), "r05_phik_pmc.json")
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
        # chip with resident wavefronts of the engine's choice of lanes per agent: TWO rounds of four wavefronts per SIMD
        # (2 x 4 x 1024 x 64 / lanes agents; one round is 1.5 - 2 % slower: the tail of a launch is a larger share)
        dict(name="configs[0]", model="omni", K=5, dt=0.1, horizon=0.5, prec="f64", bounds=MAP_BOUNDS,
             means=[[2.5, 2.5]], sigmas=[[1.5, 1.5]]),
        dict(name="configs[0], chip-filling batch", model="omni", K=5, dt=0.1, horizon=0.5, prec="f64", bounds=MAP_BOUNDS,
             means=[[2.5, 2.5]], sigmas=[[1.5, 1.5]], agents=65536),
        dict(name="configs[1]", model="simple_cart", K=10, dt=0.1, horizon=2.0, prec="f64", bounds=MAP_BOUNDS,
             means=MEANS, sigmas=SIGMAS),
        dict(name="configs[1], chip-filling batch", model="simple_cart", K=10, dt=0.1, horizon=2.0, prec="f64", bounds=MAP_BOUNDS,
             means=MEANS, sigmas=SIGMAS, agents=65536),
        dict(name="explore_omni.yaml as shipped (K = 10, T = 50)", model="omni", K=10, dt=0.1, horizon=5.0, prec="f64",
             bounds=MAP_BOUNDS, means=MEANS, sigmas=SIGMAS),
        dict(name="explore_omni.yaml as shipped, chip-filling batch", model="omni", K=10, dt=0.1, horizon=5.0, prec="f64",
             bounds=MAP_BOUNDS, means=MEANS, sigmas=SIGMAS, agents=32768),
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
    # What the two byte-lookup kernels of a tick should cost ("state the bound"): DESIGN.md section 4.5 --
    # dwa_control_kernel<MAP, FLEET> is instruction-issue bound (~245 instructions per rollout step before round 6's fast trig:
    # sincos, wrap, the two exact divisions of world2Grid, one L2-resident byte, the objective), never bandwidth bound (<= 9.8 M
    # byte lookups per tick of a 33 KB map); inflate_kernel: < 1 MB of scattered byte stores, launch + two barriers.
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
                           "bounds": {"dwa_window_lookups_per_tick_at_most": lookups, "dilation_launch_floor_us": 5.0,
                                      "note": "dwa_control_kernel<MAP, FLEET>: instruction-issue bound (20 dependent rollout steps per "
                                              "lane: sin/cos, wrap, two exact fp64 divisions of world2Grid, one L2-resident byte, the "
                                              "objective), never a bandwidth problem (the map is 33 KB); inflate_kernel: < 1 MB of "
                                              "scattered byte stores, launch + two barriers ~5 us (DESIGN.md section 4.5)"},
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
    # the same consensus on the several-agents-per-wavefront kernel (explore_omni.yaml: T = 50, omni) at a chip-filling batch, one sum
    # record per WAVEFRONT (eea_batch_io::rec_per_wavefront) against one per agent: at short horizons the record sum sets the pass time
    yaml_env = {"CONSENSUS_BENCH_HORIZON": "5.0", "CONSENSUS_BENCH_MODEL": "omni"}
    for wave in ("1", "0"):
        cases.append(("explore_omni.yaml (T = 50, omni) at 32768 agents, gated, local, sum records per %s" % ("wavefront" if wave == "1" else "agent"),
                      "", 2, "32", dict(yaml_env, CONSENSUS_BENCH_WAVE_RECORDS=wave), 32768))
    for case in cases:
        name, lib, lag = case[:3]
        mode = case[3] if len(case) > 3 else "2"
        env = dict(os.environ, **case[4]) if len(case) > 4 else None
        n_agents = case[5] if len(case) > 5 else agents
        try:
            r = subprocess.run([exe, "3000", str(n_agents), "1", lib, str(lag), mode], capture_output=True, text=True, timeout=120, env=env)
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
