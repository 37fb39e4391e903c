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


from bench_legs import (HBM_PEAK_GBS, MAP_BOUNDS, MEANS, SIGMAS, VALU_F32_PEAK_TF, VALU_F64_PEAK_TF, cpp_host_loop_leg,  # noqa: E402,F401
                        cpp_tick_latency, cpu_baseline, cpu_ticks, host_info, other_config_legs, phik_legs, single_robot_ticks,
                        tick_legs)


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
        for name in ("r06_control_pmc.json", "r06_control_pmc_spl1.json", "r05_control_pmc.json", "r05_control_pmc_spl1.json"):
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
            prof_name = "r06_bench_profile.json" if os.path.exists(os.path.join(ROOT, "profiles", "r06_bench_profile.json")) \
                else "r05_bench_profile.json"
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
            cpath = os.path.join(ROOT, "profiles", "r06_isa_counts.json")
            if not os.path.exists(cpath):
                cpath = os.path.join(ROOT, "profiles", "r05_isa_counts.json")
            if os.path.exists(cpath):
                with open(cpath) as f:
                    ic = json.load(f)
                if ic.get("T") == T and ic.get("K") == K and ic.get("precision") == args.precision and ic.get("matrix_insts_per_wave"):
                    valu, mfma, src = ic["valu_insts_per_wave_incl_matrix"], ic["matrix_insts_per_wave"], "profiles/%s (SQ counters)" % os.path.basename(cpath)
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
        gave_up = threading.Event()
        t_exchange = time.time()

        def watchdog():
            if finished.wait(limit):
                return
            gave_up.set()
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
            # an exception at the time limit is the time-out seen from the other side (a peer's watchdog ended its process and
            # the collective library reports a reset connection): the watchdog of THIS process owns the line and the exit code
            if gave_up.is_set() or time.time() - t_exchange >= limit:
                gave_up.wait(10.0)
                while gave_up.is_set():
                    time.sleep(1.0)
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
