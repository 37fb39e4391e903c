"""N > 1 path on CPU: world_size-2 gloo processes shard an agent batch, each computes its agents'
c_k (with the CPU oracle standing in for the device kernel, which is the checker's only role here)
and all-gathers them; the result must equal the single-process batch in agent order."""
import os
import socket

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist
import torch.multiprocessing as mp

from ergodic_exploration_amd import agent_batch as ab


def test_shard_ranges_cover_batch():
    for n, w in [(4096, 8), (4096, 1), (10, 4), (7, 2), (3, 8)]:
        ranges = [ab.shard_range(n, r, w) for r in range(w)]
        assert ranges[0][0] == 0 and ranges[-1][1] == n
        for a, b in zip(ranges[:-1], ranges[1:]):
            assert a[1] == b[0]
        assert sum(ab.shard_sizes(n, w)) == n
    with pytest.raises(ValueError):
        ab.shard_range(4, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _ck_for(agent_ids):
    from oracle import pyoracle as po
    K, lx, ly, T = 5, 12.0, 6.0, 20
    out = np.empty((len(agent_ids), K * K))
    for i, a in enumerate(agent_ids):
        rng = np.random.default_rng(1000 + a)
        xt = np.vstack([rng.uniform(0, lx, T), rng.uniform(0, ly, T), np.zeros(T)])
        out[i] = po.traj_coeff(lx, ly, K, xt)
    return out


def _worker(rank, world, port, n_agents, ragged, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    first, last = ab.shard_range(n_agents, rank, world)
    ck_local = torch.as_tensor(_ck_for(range(first, last)))
    if ragged:
        ck_all = ab.gather_ck_ragged(ck_local, n_agents)
    else:
        ck_all, _ = ab.gather_ck(ck_local)
        # async form used by bench.py
        out2, work = ab.gather_ck(ck_local, async_op=True)
        work.wait()
        assert torch.equal(out2, ck_all)
    mean = ab.consensus_ck(ck_all)
    # the consensus without the gather: one all-reduce of K^2 + 1 reals (unequal shards included)
    mean2 = ab.consensus_ck_allreduce(ck_local)
    assert torch.allclose(mean2, mean, rtol=0, atol=1e-15)
    if rank == 0:
        q.put((ck_all.numpy(), mean.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_agents,ragged", [(8, False), (7, True)])
def test_gather_ck_world2_gloo(n_agents, ragged):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_agents, ragged, q)) for r in range(world)]
    for p in procs:
        p.start()
    ck_all, mean = q.get(timeout=60)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    ref = _ck_for(range(n_agents))
    assert np.array_equal(ck_all, ref)
    assert np.allclose(mean, ref.mean(0), atol=1e-15)


def _phik_partial_cpu(phi, nx, ny, row0, nrows, K, res):
    """oracle: partial phi_k of a row tile (sum over its points only)"""
    from oracle import pyoracle as po
    lx, ly = (nx - 1) * res, (ny - 1) * res
    g = po.phi_grid(nx, ny, res)
    sl = slice(row0 * nx, (row0 + nrows) * nx)
    return po.spatial_coeff(lx, ly, K, phi[sl], g[:, sl])


def _tile_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    nx, ny, K, res = 33, 21, 6, 0.1
    phi = np.random.default_rng(4).random(nx * ny)          # un-normalised target values
    row0, nrows = ab.grid_row_tile(ny, rank, world)
    part = torch.as_tensor(_phik_partial_cpu(phi, nx, ny, row0, nrows, K, res))
    mass = torch.as_tensor(phi[row0 * nx:(row0 + nrows) * nx].sum())
    pk = ab.reduce_phik(part, total_mass=mass)
    pk_raw = ab.reduce_phik(part)
    if rank == 0:
        q.put((pk.numpy(), pk_raw.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_grid_tiled_phik_world2_gloo():
    from oracle import pyoracle as po
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_tile_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    pk, pk_raw = q.get(timeout=60)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    nx, ny, K, res = 33, 21, 6, 0.1
    phi = np.random.default_rng(4).random(nx * ny)
    g = po.phi_grid(nx, ny, res)
    full = po.spatial_coeff((nx - 1) * res, (ny - 1) * res, K, phi, g)
    assert np.abs(pk_raw - full).max() < 1e-12
    assert np.abs(pk - full / phi.sum()).max() < 1e-13
    rows = [ab.grid_row_tile(ny, r, 3) for r in range(3)]
    assert rows[0][0] == 0 and sum(n for _, n in rows) == ny


def _occ_tile_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import pyoracle as po
    nx, ny, K, res = 40, 27, 5, 0.1
    occ = np.random.default_rng(9).choice(np.array([0, 100, -1, 37], dtype=np.int8), size=nx * ny)
    ent = np.array([po.lib().eo_entropy(float(v) / 100.0) for v in occ])   # un-normalised target
    row0, nrows = ab.grid_row_tile(ny, rank, world)
    part = torch.as_tensor(_phik_partial_cpu(ent, nx, ny, row0, nrows, K, res))
    pk = ab.reduce_occupancy_sums(part)
    if rank == 0:
        q.put(pk.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_occupancy_tiled_phik_world2_gloo():
    """row-tiled occupancy target (BASELINE config 5 sharding): the all-reduced un-normalised sums
    divided by their (0,0) element equal spatialCoeff of the normalised entropy target"""
    from oracle import pyoracle as po
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_occ_tile_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    pk = q.get(timeout=60)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    nx, ny, K, res = 40, 27, 5, 0.1
    occ = np.random.default_rng(9).choice(np.array([0, 100, -1, 37], dtype=np.int8), size=nx * ny)
    ent = np.array([po.lib().eo_entropy(float(v) / 100.0) for v in occ])
    full = po.spatial_coeff((nx - 1) * res, (ny - 1) * res, K, ent / ent.sum(), po.phi_grid(nx, ny, res))
    assert np.abs(pk - full).max() < 1e-13
    assert abs(pk[0] - 1.0) < 1e-15


def test_bench_starts_its_own_ranks_dry_run(tmp_path):
    """`python bench.py --gpus 2 --steps K --warmup W` with no launcher around it (what the driver types): the
    parent -- which must not touch the GPU -- starts two ranks through torch.distributed.run and exits with their
    return code; EEA_BENCH_DRYRUN keeps the ranks off the device so that the plumbing is testable here.  (The
    real thing runs on the GPU box: tests/test_gpu_round2_parity.py::test_bench_self_launches_its_ranks.)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, EEA_BENCH_DRYRUN="1", EEA_BENCH_DETAIL_DIR=str(tmp_path))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    gt = rec.pop("grid_tile")
    assert rec.pop("detail") == "bench_detail.json"
    assert rec == {"dryrun": True, "n_gpus": 2, "gpus_arg": 2, "steps": 3, "warmup": 1}
    # the grid-tile leg's partition + all-reduce + normalisation (numpy in place of the device kernel): 48 rows over 2 ranks
    assert gt["ok"] and gt["rows_per_rank"] == 24 and gt["max_abs_err_vs_untiled"] < 1e-12
    # a failing rank must surface as a non-zero exit code of the parent
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--bogus-flag"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
