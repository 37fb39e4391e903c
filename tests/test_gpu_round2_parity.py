"""GPU parity cases added in round 2 (all through the C ABI, checker = the CPU oracle):

* the decentralised-consensus input (eea_batch_io::d_ck_shared) against the oracle's shared-c_k switch;
* the exchange entry points (eea_ck_sum, eea_comm_*) on a single-rank RCCL communicator;
* BASELINE config 4 at FULL size: 4096 agents, K = 10, T = 200, u0 and the whole warm-start matrix ut
  against the oracle run on every host thread;
* BASELINE config 5 END TO END: 1024 x 1024 seed-2024 occupancy grid -> eea_set_target_occupancy -> K = 30,
  T = 500 control on the 102.4 m domain, phi_k and two consecutive control calls against the oracle;
* `python bench.py --gpus 2` started the way the driver starts it (no launcher around it).

Tolerances: c_k, phi_k <= 1e-11; trajectory / co-state / controls <= 1e-9.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from oracle import pyoracle as po
from ergodic_exploration_amd import capi
from tests.gpu_util import MAP_BOUNDS, MEANS, SIGMAS, MODELS, angle_diff, make_pair, random_poses

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL_CK = 1e-11
TOL = 1e-9


def dev(a, dtype=torch.float64):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).cuda()


@pytest.mark.parametrize("model,K,horizon,n_mem", [("simple_cart", 10, 20.0, 0), ("omni", 10, 5.0, 7),
                                                   ("omni", 7, 3.0, 0), ("omni", 20, 5.0, 0)])
def test_consensus_shared_ck_against_oracle(model, K, horizon, n_mem):
    """d_ck_shared replaces the agent's own c_k in fourier_diff (ergodic_control.hpp:422); the own c_k is
    still what d_ck receives.  c_bar = mean of the agents' c_k of the previous call, as bench.py feeds it."""
    rng = np.random.default_rng(42)
    B = 5
    eng, ors = make_pair(model, K, horizon, n_oracles=B)
    T, K2 = eng.T, eng.K2
    poses = random_poses(rng, B)
    ut0 = rng.uniform(-0.5, 0.5, (B, T, 3))
    if model == "simple_cart":
        ut0[:, :, 1] = 0.0
    mem = random_poses(rng, B * n_mem).reshape(B, n_mem, 3) if n_mem else None
    d_pose, d_ut = dev(poses), dev(ut0)
    d_u0 = torch.empty((B, 3), dtype=torch.float64, device="cuda")
    d_ck = torch.empty((B, K2), dtype=torch.float64, device="cuda")
    outs = {k: torch.empty((B, T, 3), dtype=torch.float64, device="cuda") for k in ("edx", "rhot")}
    d_mem = dev(mem) if n_mem else None
    d_nmem = torch.full((B,), n_mem, dtype=torch.int32, device="cuda") if n_mem else None
    sums = torch.empty((K2 + 1,), dtype=torch.float64, device="cuda")
    d_cbar = torch.empty((K2,), dtype=torch.float64, device="cuda")
    comm = capi.Comm(0, 1, 0, None)
    for b in range(B):
        ors[b].ut = ut0[b].T
    cbar = None
    for call in range(3):
        eng.control_batch(B, d_pose, d_ut, d_u0, mem_cols=d_mem, n_mem=d_nmem, mem_stride=n_mem, ck=d_ck,
                          ck_shared=(d_cbar if cbar is not None else None), **outs)
        torch.cuda.synchronize()
        ck = d_ck.cpu().numpy()
        for b in range(B):
            ors[b].set_shared_ck(cbar)
            u, st = ors[b].control(MAP_BOUNDS, poses[b], mem[b].T if n_mem else None, stages=True)
            assert np.abs(ck[b] - st["ck"]).max() <= TOL_CK          # own c_k, unaffected by the switch
            assert np.abs(outs["edx"][b].cpu().numpy().T - st["edx"]).max() <= TOL
            assert np.abs(outs["rhot"][b].cpu().numpy().T - st["rhot"]).max() <= TOL
            assert np.abs(d_ut[b].cpu().numpy().T - st["ut"]).max() <= TOL
            assert np.abs(d_u0[b].cpu().numpy() - u).max() <= TOL
            ors[b].ut = d_ut[b].cpu().numpy().T
        # consensus of this call's c_k through the exchange entry points (local communicator)
        eng.ck_sum(B, d_ck, sums)
        comm.consensus_ck(eng, B, d_ck, d_cbar)
        torch.cuda.synchronize()
        s = sums.cpu().numpy()
        assert s[K2] == B and np.abs(s[:K2] - ck.sum(0)).max() < 1e-13
        cbar = d_cbar.cpu().numpy()
        assert np.abs(cbar - ck.mean(0)).max() < 1e-14
    # with the consensus equal to the own c_k the result is the reference behaviour again
    eng.close()
    comm.close()


def test_exchange_steps_on_single_rank_rccl():
    """eea_comm_* with a real RCCL communicator of one rank (all a 1-GPU box can hold): the all-gather
    returns the rank's c_k, the all-reduce leaves sums unchanged, the consensus is the local mean."""
    eng, _ = make_pair("omni", 10, 2.0, n_oracles=0)
    K2, B = eng.K2, 37
    rng = np.random.default_rng(3)
    ck = dev(rng.normal(size=(B, K2)))
    comm = capi.Comm(0, 1, 0, capi.comm_unique_id())
    allc = torch.zeros((B, K2), dtype=torch.float64, device="cuda")
    cbar = torch.zeros((K2,), dtype=torch.float64, device="cuda")
    buf = ck[0].clone()
    s = torch.cuda.Stream()
    comm.allgather_ck(eng, B, ck, allc, stream=s.cuda_stream)
    comm.consensus_ck(eng, B, ck, cbar, stream=s.cuda_stream)
    comm.allreduce_sum(eng, buf, K2, stream=s.cuda_stream)
    torch.cuda.synchronize()
    assert torch.equal(allc, ck)
    assert torch.equal(buf, ck[0])
    assert np.abs(cbar.cpu().numpy() - ck.cpu().numpy().mean(0)).max() < 1e-14
    # asynchronous forms: ordered after the compute stream, completion per slot
    cbar2 = torch.zeros((K2,), dtype=torch.float64, device="cuda")
    all2 = torch.zeros((B, K2), dtype=torch.float64, device="cuda")
    with torch.cuda.stream(s):
        ck2 = ck * 2.0          # produced on the compute stream right before the exchange starts
    comm.consensus_ck_async(eng, B, ck2, cbar2, s.cuda_stream, 1)
    comm.allgather_ck_async(eng, B, ck2, all2, s.cuda_stream, 2)
    comm.wait(1, s.cuda_stream)
    comm.wait(2, s.cuda_stream)
    comm.wait(3, s.cuda_stream)   # never started: no-op
    with torch.cuda.stream(s):
        got = cbar2.clone()
    torch.cuda.synchronize()
    assert np.abs(got.cpu().numpy() - 2.0 * ck.cpu().numpy().mean(0)).max() < 1e-13
    assert torch.equal(all2, ck2)
    # fp32 engine: same entry points on float buffers
    e32, _ = make_pair("omni", 10, 2.0, n_oracles=0, precision=capi.PREC_F32)
    ck32 = ck.float()
    cbar32 = torch.zeros((K2,), dtype=torch.float32, device="cuda")
    comm.consensus_ck(e32, B, ck32, cbar32)
    torch.cuda.synchronize()
    assert np.abs(cbar32.cpu().numpy() - ck32.cpu().numpy().astype(np.float64).mean(0)).max() < 1e-5
    comm.close()
    eng.close()
    e32.close()


def _sum_record_case(model, K, horizon, B, precision=capi.PREC_F64, n_mem=0, bad=(), dt=0.1):
    """one control_batch with d_ck and d_ck_rec, then eea_ck_records_sum: (sum record, per-agent c_k, status) as numpy"""
    rng = np.random.default_rng(100 + B + K)
    eng, _ = make_pair(model, K, horizon, n_oracles=0, precision=precision, dt=dt)
    T, K2 = eng.T, eng.K2
    tdt = torch.float64 if precision == capi.PREC_F64 else torch.float32
    poses = random_poses(rng, B)
    ut0 = rng.uniform(-0.5, 0.5, (B, T, 3))
    if model == "simple_cart":
        ut0[:, :, 1] = 0.0
        for b in bad:
            ut0[b, T // 2, 1] = 0.3      # a lateral velocity: SimpleCart::operator() throws (cart.hpp:167-170)
    mem = random_poses(rng, B * n_mem).reshape(B, n_mem, 3) if n_mem else None
    d_pose, d_ut0 = dev(poses, tdt), dev(ut0, tdt)
    d_u0 = torch.empty((B, 3), dtype=tdt, device="cuda")
    d_ck = torch.zeros((B, K2), dtype=tdt, device="cuda")
    d_status = torch.full((B,), -1, dtype=torch.int32, device="cuda")
    d_mem = dev(mem, tdt) if n_mem else None
    d_nmem = torch.full((B,), n_mem, dtype=torch.int32, device="cuda") if n_mem else None
    L = eng.ck_record_len
    assert L == ((K2 + 2) & ~1)
    recs = []
    for rep in range(2):   # twice from the same inputs: the record is run-to-run deterministic (fixed summation order)
        d_ut = d_ut0.clone()
        rec = torch.full((L,), float("nan"), dtype=tdt, device="cuda")
        arec = torch.full((B, L), float("nan"), dtype=tdt, device="cuda")
        eng.control_batch(B, d_pose, d_ut, d_u0, mem_cols=d_mem, n_mem=d_nmem, mem_stride=n_mem, ck=d_ck,
                          status=d_status, ck_rec=arec)
        eng.ck_records_sum(B, arec, rec)
        torch.cuda.synchronize()
        recs.append(rec.cpu().numpy().astype(np.float64))
        # the per-agent records: [own c_k, 1, pad] -- all zero for an agent the model rejects
        a = arec.cpu().numpy().astype(np.float64)
        ok = d_status.cpu().numpy() == 0
        assert np.array_equal(a[ok, :K2], d_ck.cpu().numpy().astype(np.float64)[ok]) and (a[ok, K2] == 1).all()
        assert (a[:, K2 + 1:] == 0).all() and (a[~ok] == 0).all()
    assert np.array_equal(recs[0], recs[1])
    out = recs[0], d_ck.cpu().numpy().astype(np.float64), d_status.cpu().numpy(), K2
    eng.close()
    return out


@pytest.mark.parametrize("model,K,horizon,B", [
    ("simple_cart", 10, 20.0, 4096 + 3),   # the metric shape, 65 groups (the last one with 3 agents) + the ticket
    ("omni", 10, 20.0, 64),                # exactly one group: no ticket
    ("omni", 10, 5.0, 1),
    ("omni", 5, 0.5, 130),                 # K = 5 (block contraction, one record element per lane)
    ("omni", 12, 3.0, 200),                # generic K <= 16 instance
    ("omni", 20, 5.0, 77),                 # K = 20: 402-element record, single-agent workgroups
    ("omni", 30, 6.0, 150),                # workgroup-per-agent kernel
])
def test_sum_record_of_the_launch(model, K, horizon, B):
    """eea_batch_io::d_ck_rec + eea_ck_records_sum (ABI 3): per-agent records out of the control kernel, ONE small
    launch adds them to [sum_b c_k, number of agents, pad].  Against the per-agent c_k of the same call (d_ck):
    summation order differs, so <= 1e-12 relative to the sum; count exact; pad zero; bitwise reproducible."""
    rec, ck, status, K2 = _sum_record_case(model, K, horizon, B)
    assert (status == 0).all()
    assert rec[K2] == B and (rec[K2 + 1:] == 0).all()
    ref = ck.sum(0)
    assert np.abs(rec[:K2] - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("K,B", [(10, 1), (10, 32), (10, 33), (10, 256), (10, 257), (10, 288), (10, 2049), (10, 8200),
                                 (5, 300), (20, 517), (3, 97)])
def test_record_sum_tree_order(K, B):
    """eea_ck_records_sum on synthetic records against the documented summation tree taken literally in numpy
    (include/ergodic_amd.h: groups of 32 agents in agent order, 8 group records per level-1 record, the level-1 records
    in order): BITWISE, for the tree's edge shapes -- one group, a level-1 group of one member (257 ... 288 agents), more
    than 8 level-1 records (8200 agents), records that end inside an element slice of 64 (K = 5, 20, 3)."""
    lim = np.array([1.0, 1.0, 2.0])
    eng = capi.Engine(capi.make_config(capi.MODEL_OMNI, 0.1, 1.0, 0.1, 1.0, K, np.diag([1.0, 1.0, 2.0]), -lim, lim))
    L = eng.ck_record_len
    rng = np.random.default_rng(7 * B + K)
    a = rng.standard_normal((B, L)) * 10.0 ** rng.integers(-3, 4, (B, 1))   # magnitudes that make the order matter
    d_a = torch.as_tensor(a).cuda()
    out = torch.full((L,), float("nan"), dtype=torch.float64, device="cuda")
    for _ in range(3):   # the tickets reset themselves: back-to-back launches into the same output
        eng.ck_records_sum(B, d_a, out)
    torch.cuda.synchronize()
    got = out.cpu().numpy()

    def seq_sum(rows):
        acc = np.zeros(L)
        for r in rows:
            acc = acc + r
        return acc
    g0 = [seq_sum(a[i:i + 32]) for i in range(0, B, 32)]
    if len(g0) == 1:
        ref = g0[0]
    else:
        g1 = [seq_sum(g0[i:i + 8]) for i in range(0, len(g0), 8)]
        ref = g1[0] if len(g1) == 1 else seq_sum(g1)
    assert np.array_equal(got, ref)
    assert B <= 64 or not np.array_equal(ref, seq_sum(a))   # (the order does matter for these inputs)
    eng.close()


def test_sum_record_skips_rejected_agents_and_fp32():
    """agents SimpleCart rejects (EEA_ERR_INVALID_TWIST) contribute nothing and are not counted; fp32 engine"""
    bad = (0, 5, 63, 64, 199)
    rec, ck, status, K2 = _sum_record_case("simple_cart", 10, 20.0, 200, bad=bad)
    assert sorted(np.nonzero(status == 2)[0]) == list(bad)
    good = status == 0
    assert rec[K2] == 200 - len(bad)
    assert np.abs(rec[:K2] - ck[good].sum(0)).max() <= 1e-12 * np.abs(ck[good].sum(0)).max()
    rec, ck, status, K2 = _sum_record_case("omni", 10, 20.0, 300, precision=capi.PREC_F32, n_mem=40)
    assert rec[K2] == 300
    assert np.abs(rec[:K2] - ck.sum(0)).max() <= 2e-5 * np.abs(ck.sum(0)).max()


@pytest.mark.parametrize("K,horizon", [(10, 20.0), (30, 6.0)])
def test_shared_ck_as_sum_records(K, horizon):
    """ck_shared_parts = n (ABI 3): d_ck_shared holds n sum records and the kernel uses sum of sums x (1 / sum of counts) -- the
    reciprocal is formed once per wavefront, round 6 -- bitwise the same controls as the ABI-2 form fed with that product, for both
    control kernels."""
    rng = np.random.default_rng(8)
    B = 150
    eng, _ = make_pair("omni", K, horizon, n_oracles=0)
    T, K2, L = eng.T, eng.K2, eng.ck_record_len
    poses = random_poses(rng, B)
    ut0 = rng.uniform(-0.5, 0.5, (B, T, 3))
    d_pose, d_u0 = dev(poses), torch.empty((B, 3), dtype=torch.float64, device="cuda")
    half = 70
    recs = torch.zeros((2, L), dtype=torch.float64, device="cuda")
    arec = torch.zeros((B, L), dtype=torch.float64, device="cuda")
    d_ut = dev(ut0)
    # pass 1: two agent groups, the records of each group added separately
    eng.control_batch(half, d_pose[:half], d_ut[:half], d_u0[:half], ck_rec=arec[:half])
    eng.control_batch(B - half, d_pose[half:], d_ut[half:], d_u0[half:], ck_rec=arec[half:])
    eng.ck_records_sum(half, arec[:half], recs[0])
    eng.ck_records_sum(B - half, arec[half:], recs[1])
    torch.cuda.synchronize()
    r = recs.cpu().numpy()
    assert r[0, K2] == half and r[1, K2] == B - half
    cbar = (r[0, :K2] + r[1, :K2]) * (1.0 / (r[0, K2] + r[1, K2]))
    # pass 2 consumes the two records ...
    ut_a = d_ut.clone()
    eng.control_batch(B, d_pose, ut_a, d_u0, ck_shared=recs, ck_shared_parts=2)
    torch.cuda.synchronize()
    u_a = d_u0.cpu().numpy().copy()
    # ... or the quotient formed on the host (ABI-2 form)
    ut_b = d_ut.clone()
    eng.control_batch(B, d_pose, ut_b, d_u0, ck_shared=dev(cbar))
    torch.cuda.synchronize()
    assert np.array_equal(u_a, d_u0.cpu().numpy()) and torch.equal(ut_a, ut_b)
    assert np.isfinite(u_a).all()
    eng.close()


def _consensus_reference(eng, B, gb, d_pose, ut0, lag, passes, L):
    """the consensus passes issued one call at a time on one stream with explicit synchronisation: pass i consumes the
    sum record of pass i - lag (ck_shared_parts = 1), two agent groups"""
    ut, u0 = dev(ut0), torch.empty((B, 3), dtype=torch.float64, device="cuda")
    arec = torch.zeros((B, L), dtype=torch.float64, device="cuda")
    sums = [torch.zeros((L,), dtype=torch.float64, device="cuda") for _ in range(passes)]
    for i in range(passes):
        src = i - lag
        for g in range(len(gb) - 1):
            sl = slice(gb[g], gb[g + 1])
            eng.control_batch(gb[g + 1] - gb[g], d_pose[sl], ut[sl], u0[sl], ck_rec=arec[sl],
                              ck_shared=sums[src] if src >= 0 else None, ck_shared_parts=1 if src >= 0 else 0)
        torch.cuda.synchronize()
        eng.ck_records_sum(B, arec, sums[i])
        torch.cuda.synchronize()
    return ut, u0, sums


@pytest.mark.parametrize("lag", [1, 3])
@pytest.mark.parametrize("rccl", [False, True])
def test_consensus_pass_stream_ordered_exchange(lag, rccl):
    """eea_comm_records_exchange_async (events on the group streams, record sum + all-reduce on the communicator's
    stream) + eea_comm_wait on the consuming streams, no host synchronisation in between -- bitwise the synchronised
    sequence after 7 passes; over the local communicator and over a REAL one-rank RCCL communicator (the collective
    branch of the exchange: ncclAllReduce of the 816-byte record)."""
    rng = np.random.default_rng(17)
    B, K, G, NB, passes = 300, 10, 2, 6, 7
    eng, _ = make_pair("omni", K, 20.0, n_oracles=0)
    T, K2, L = eng.T, eng.K2, eng.ck_record_len
    d_pose = dev(random_poses(rng, B))
    ut0 = rng.uniform(-0.3, 0.3, (B, T, 3))
    gb = [0, 130, B]
    ut_a, u0_a, sums_a = _consensus_reference(eng, B, gb, d_pose, ut0, lag, passes, L)
    comm = capi.Comm(0, 1, 0, capi.comm_unique_id() if rccl else None)
    streams = [torch.cuda.Stream() for _ in range(G)]
    ut_b, u0_b = dev(ut0), torch.empty((B, 3), dtype=torch.float64, device="cuda")
    arecs = [torch.zeros((B, L), dtype=torch.float64, device="cuda") for _ in range(NB)]
    sums_b = [torch.zeros((L,), dtype=torch.float64, device="cuda") for _ in range(NB)]
    torch.cuda.synchronize()
    for i in range(passes):
        slot, src = i % NB, (i - lag) % NB if i >= lag else None
        for g in range(G):
            sl = slice(gb[g], gb[g + 1])
            if src is not None:
                comm.wait(src, streams[g].cuda_stream)
            eng.control_batch(gb[g + 1] - gb[g], d_pose[sl], ut_b[sl], u0_b[sl], ck_rec=arecs[slot][sl],
                              ck_shared=None if src is None else sums_b[src], ck_shared_parts=0 if src is None else 1,
                              stream=streams[g].cuda_stream)
        comm.records_exchange_async(eng, B, arecs[slot], sums_b[slot], [st.cuda_stream for st in streams], slot)
    torch.cuda.synchronize()
    assert torch.equal(ut_a, ut_b) and torch.equal(u0_a, u0_b)
    assert torch.equal(sums_a[passes - 1], sums_b[(passes - 1) % NB])
    assert float(sums_a[passes - 1][K2]) == B
    comm.close()
    eng.close()


@pytest.mark.parametrize("lag", [1, 2, 4])
@pytest.mark.parametrize("rccl", [False, True])
@pytest.mark.parametrize("K,horizon", [(10, 20.0), (30, 6.0)])
def test_consensus_pass_device_bound_exchange(lag, rccl, K, horizon):
    """The DEVICE-BOUND exchange (ABI 4), as bench.py's consensus leg issues it: per pass the two groups' control calls
    (eea_batch_io::d_rec_ready / rec_seq out, d_ck_flag / ck_flag_seq in) and ONE eea_comm_records_exchange_bound -- no
    host wait, no stream wait, no event anywhere: the record sum polls the agents' ready marks, the consuming kernels poll
    the exchange's flag right before the first use of c_bar.  Bitwise the synchronised sequence after 9 passes, lag 1
    (the previous pass's c_bar), 2 and 4; local communicator and a real one-rank RCCL communicator (sum -> ncclAllReduce
    -> publish); the wavefront kernel and the workgroup kernel (K = 30).  No agent may report a timeout."""
    rng = np.random.default_rng(23)
    B, G, NB, passes = 300, 2, 6, 9
    eng, _ = make_pair("omni", K, horizon, n_oracles=0)
    T, K2, L = eng.T, eng.K2, eng.ck_record_len
    d_pose = dev(random_poses(rng, B))
    ut0 = rng.uniform(-0.3, 0.3, (B, T, 3))
    gb = [0, 130, B]
    ut_a, u0_a, sums_a = _consensus_reference(eng, B, gb, d_pose, ut0, lag, passes, L)
    comm = capi.Comm(0, 1, 0, capi.comm_unique_id() if rccl else None)
    streams = [torch.cuda.Stream() for _ in range(G)]
    ut_b, u0_b = dev(ut0), torch.empty((B, 3), dtype=torch.float64, device="cuda")
    arecs = [torch.zeros((B, L), dtype=torch.float64, device="cuda") for _ in range(NB)]
    sums_b = [torch.zeros((L,), dtype=torch.float64, device="cuda") for _ in range(NB)]
    ready = torch.zeros((B,), dtype=torch.int32, device="cuda")
    flag = torch.zeros((1,), dtype=torch.int32, device="cuda")
    status = torch.full((B,), -1, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    for i in range(passes):
        seq = i + 1
        slot, src = i % NB, (i - lag) % NB if i >= lag else None
        for g in range(G):
            sl = slice(gb[g], gb[g + 1])
            eng.control_batch(gb[g + 1] - gb[g], d_pose[sl], ut_b[sl], u0_b[sl], ck_rec=arecs[slot][sl],
                              rec_ready=ready[sl], rec_seq=seq, status=status[sl],
                              ck_shared=None if src is None else sums_b[src], ck_shared_parts=0 if src is None else 1,
                              ck_flag=None if src is None else flag, ck_flag_seq=seq - lag,
                              stream=streams[g].cuda_stream)
        comm.records_exchange_bound(eng, B, arecs[slot], ready, seq, sums_b[slot], flag, slot)
    torch.cuda.synchronize()
    assert (status.cpu().numpy() == 0).all()
    assert int(flag.item()) == passes and (ready.cpu().numpy() == passes).all()
    assert torch.equal(ut_a, ut_b) and torch.equal(u0_a, u0_b)
    assert torch.equal(sums_a[passes - 1], sums_b[(passes - 1) % NB])
    assert float(sums_a[passes - 1][K2]) == B
    comm.close()
    eng.close()


def test_device_bound_exchange_is_one_step_per_launch():
    """eea_control_batch_steps refuses the device-bound exchange fields with n_steps > 1: a step of the launch would wait,
    inside the kernel, for an exchange the host can only enqueue AFTER the launch -- that needs truly concurrent hardware
    queues, and streams may share one (measured in round 4: the same test passed or ran into the time-out depending on
    which streams the process had created before).  Every wait of the protocol is for work enqueued BEFORE the waiter."""
    eng, _ = make_pair("omni", 10, 20.0, n_oracles=0)
    B, T, L = 8, eng.T, eng.ck_record_len
    d_pose = dev(random_poses(np.random.default_rng(0), B))
    ut = torch.zeros((B, T, 3), dtype=torch.float64, device="cuda")
    u0 = torch.empty((B, 3), dtype=torch.float64, device="cuda")
    arec = torch.zeros((B, L), dtype=torch.float64, device="cuda")
    ready = torch.zeros((B,), dtype=torch.int32, device="cuda")
    flag = torch.zeros((1,), dtype=torch.int32, device="cuda")
    rec = torch.zeros((L,), dtype=torch.float64, device="cuda")
    for kw in (dict(ck_rec=arec, rec_ready=ready, rec_seq=1), dict(ck_shared=rec, ck_shared_parts=1, ck_flag=flag, ck_flag_seq=0)):
        with pytest.raises(capi.EngineError) as ei:
            eng.control_batch(B, d_pose, ut, u0, n_steps=3, **kw)
        assert ei.value.status == capi.ERR_UNSUPPORTED
    # (plain records / a plain shared c_k are fine with several steps per launch)
    eng.control_batch(B, d_pose, ut, u0, n_steps=3, ck_rec=arec, ck_shared=rec, ck_shared_parts=1)
    torch.cuda.synchronize()
    assert torch.isfinite(ut).all()
    eng.close()


def test_consensus_loop_needs_no_host_cores():
    """the per-pass consensus loop must not depend on spinning host threads.  The device-bound
    exchange has none (the calling thread issues three launches per pass and waits for nothing), so the whole loop --
    4096 agents, two groups, lag 1, 300 passes -- is run in a child process pinned to ONE CPU (sched_setaffinity before
    anything else starts, inherited by every runtime thread): it must finish, with no agent timed out, at a pass time of
    the same order as unpinned."""
    code = r"""
import os, sys, time
os.sched_setaffinity(0, {sorted(os.sched_getaffinity(0))[0]})
sys.path.insert(0, %r)
import numpy as np, torch
from ergodic_exploration_amd import capi
from tests.gpu_util import make_pair, random_poses
rng = np.random.default_rng(5)
B, G, NB, passes, lag = 4096, 2, 4, 300, 1
eng, _ = make_pair("simple_cart", 10, 20.0, n_oracles=0)
T, L = eng.T, eng.ck_record_len
d_pose = torch.as_tensor(random_poses(rng, B)).cuda()
ut = torch.zeros((B, T, 3), dtype=torch.float64, device="cuda")
u0 = torch.empty((B, 3), dtype=torch.float64, device="cuda")
arecs = [torch.zeros((B, L), dtype=torch.float64, device="cuda") for _ in range(NB)]
sums = [torch.zeros((L,), dtype=torch.float64, device="cuda") for _ in range(NB)]
ready = torch.zeros((B,), dtype=torch.int32, device="cuda")
flag = torch.zeros((1,), dtype=torch.int32, device="cuda")
status = torch.zeros((B,), dtype=torch.int32, device="cuda")
comm = capi.Comm(0, 1, 0, None)
streams = [torch.cuda.Stream() for _ in range(G)]
gb = [0, B // 2, B]
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(passes):
    seq, slot = i + 1, i %% NB
    src = (i - lag) %% NB if i >= lag else None
    for g in range(G):
        sl = slice(gb[g], gb[g + 1])
        eng.control_batch(gb[g + 1] - gb[g], d_pose[sl], ut[sl], u0[sl], ck_rec=arecs[slot][sl], rec_ready=ready[sl],
                          rec_seq=seq, status=status[sl], ck_shared=None if src is None else sums[src],
                          ck_shared_parts=0 if src is None else 1, ck_flag=None if src is None else flag,
                          ck_flag_seq=seq - lag, stream=streams[g].cuda_stream)
    comm.records_exchange_bound(eng, B, arecs[slot], ready, seq, sums[slot], flag, slot)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("PINNED_OK cpus=%%d us_per_pass=%%.1f timeouts=%%d flag=%%d" %% (len(os.sched_getaffinity(0)), 1e6 * dt / passes,
      int((status != 0).sum().item()), int(flag.item())))
""" % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    line = [l for l in r.stdout.splitlines() if l.startswith("PINNED_OK")][-1]
    fields = dict(kv.split("=") for kv in line.split()[1:])
    assert int(fields["timeouts"]) == 0 and int(fields["flag"]) == 300
    if int(fields["cpus"]) != 1:
        # (seen on some boxes of the pool: a runtime library widens the process's affinity mask again while it loads;
        # the loop still finished without a time-out, but the claim of this test cannot be made there)
        pytest.skip("the affinity mask did not stay at one CPU on this host (%s CPUs at the end)" % fields["cpus"])
    assert float(fields["us_per_pass"]) < 500.0   # (host-bound through per-pass Python struct building; the point is: it finishes)
    if os.environ.get("EEA_PRINT_WORST"):
        print("consensus loop pinned to one CPU:", line)


def test_device_bound_exchange_times_out_instead_of_hanging():
    """A consumer whose flag never arrives, and a record sum whose producers never report, give up after about a
    second: per-agent EEA_ERR_TIMEOUT, the agent's OWN c_k in the gradient (bitwise the call without a shared c_k);
    a sum that gave up marks its record with a negative agent count, which its consumers report the same way."""
    rng = np.random.default_rng(29)
    for K, horizon in ((10, 20.0), (30, 6.0)):
        B = 6
        eng, _ = make_pair("omni", K, horizon, n_oracles=0)
        T, K2, L = eng.T, eng.K2, eng.ck_record_len
        d_pose = dev(random_poses(rng, B))
        ut0 = rng.uniform(-0.3, 0.3, (B, T, 3))
        ut_ref, u0_ref = dev(ut0), torch.empty((B, 3), dtype=torch.float64, device="cuda")
        eng.control_batch(B, d_pose, ut_ref, u0_ref)
        # (1) flag never set
        flag = torch.zeros((1,), dtype=torch.int32, device="cuda")
        rec = torch.ones((L,), dtype=torch.float64, device="cuda")
        status = torch.full((B,), -1, dtype=torch.int32, device="cuda")
        ut_a, u0_a = dev(ut0), torch.empty((B, 3), dtype=torch.float64, device="cuda")
        eng.control_batch(B, d_pose, ut_a, u0_a, ck_shared=rec, ck_shared_parts=1, ck_flag=flag, ck_flag_seq=5, status=status)
        torch.cuda.synchronize()
        assert (status.cpu().numpy() == capi.ERR_TIMEOUT).all()
        assert torch.equal(ut_a, ut_ref) and torch.equal(u0_a, u0_ref)
        # (2) producers never report: the sum gives up, publishes its flag with a negative count
        arec = torch.zeros((B, L), dtype=torch.float64, device="cuda")
        ready = torch.zeros((B,), dtype=torch.int32, device="cuda")
        out = torch.zeros((L,), dtype=torch.float64, device="cuda")
        eng.ck_records_sum_bound(B, arec, ready, 3, out, flag=flag)
        torch.cuda.synchronize()
        assert int(flag.item()) == 3 and float(out[K2]) < 0
        status.fill_(-1)
        ut_b, u0_b = dev(ut0), torch.empty((B, 3), dtype=torch.float64, device="cuda")
        eng.control_batch(B, d_pose, ut_b, u0_b, ck_shared=out, ck_shared_parts=1, ck_flag=flag, ck_flag_seq=3, status=status)
        torch.cuda.synchronize()
        assert (status.cpu().numpy() == capi.ERR_TIMEOUT).all()
        assert torch.equal(ut_b, ut_ref) and torch.equal(u0_b, u0_ref)
        eng.close()


@pytest.mark.parametrize("model", ["simple_cart", "omni"])
def test_config4_full_size_against_oracle(model):
    """BASELINE config 4 at full size: every one of the 4096 agents' u0 and warm-start matrix ut after two
    control() calls from a zero warm start against independent oracle controllers (one per host thread)."""
    B, K, horizon, dt = 4096, 10, 20.0, 0.1
    om, em, rdiag, lim = MODELS[model]
    eng, _ = make_pair(model, K, horizon, n_oracles=0)
    T = eng.T
    rng = np.random.default_rng(12345)
    poses = random_poses(rng, B)
    d_pose = dev(poses)
    d_ut = torch.zeros((B, T, 3), dtype=torch.float64, device="cuda")
    d_u0 = torch.empty((B, 3), dtype=torch.float64, device="cuda")
    limv = np.array(lim)
    cfg = po.make_config(om, dt, horizon, 0.1, 1.0, K, np.diag(rdiag), -limv, limv)
    threads = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    worst = []
    for calls in (1, 2):
        eng.control_batch(B, d_pose, d_ut, d_u0)
        torch.cuda.synchronize()
        # the helper runs one warm-up call + (calls - 1) further calls == `calls` control() calls per agent
        u_ref, ut_ref = po.batch_control(cfg, MEANS, SIGMAS, MAP_BOUNDS, poses, calls - 1, min(threads, 64))
        ut = d_ut.cpu().numpy()
        dut = np.abs(ut - ut_ref).max()
        du = np.abs(d_u0.cpu().numpy() - ut_ref[:, 0, :]).max()
        worst.append((du, dut))
        if calls > 1:
            assert np.abs(d_u0.cpu().numpy() - u_ref).max() == du
    if os.environ.get("EEA_PRINT_WORST"):
        print("config 4 full size", model, worst)
    # first call: the single-call bar; second (dependent) call starts from controls that already differ
    assert worst[0][0] <= TOL and worst[0][1] <= TOL, worst
    assert worst[1][0] <= 10 * TOL and worst[1][1] <= 10 * TOL, worst
    eng.close()


def test_config4_full_size_f32_against_f64_oracle():
    """The fp32 engine on BASELINE config 4 at full size (4096 agents, K = 10, T = 200) against the fp64 oracle:
    u <= 1e-4 on every agent's u0 and on the whole updated warm-start matrix, first call from a
    zero warm start."""
    B, K, horizon, dt = 4096, 10, 20.0, 0.1
    worst = {}
    for model in ("simple_cart", "omni"):
        om, em, rdiag, lim = MODELS[model]
        eng, _ = make_pair(model, K, horizon, n_oracles=0, precision=capi.PREC_F32)
        T = eng.T
        rng = np.random.default_rng(12345)
        poses = random_poses(rng, B)
        d_pose = dev(poses, torch.float32)
        d_ut = torch.zeros((B, T, 3), dtype=torch.float32, device="cuda")
        d_u0 = torch.empty((B, 3), dtype=torch.float32, device="cuda")
        limv = np.array(lim)
        cfg = po.make_config(om, dt, horizon, 0.1, 1.0, K, np.diag(rdiag), -limv, limv)
        threads = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        eng.control_batch(B, d_pose, d_ut, d_u0)
        torch.cuda.synchronize()
        # the oracle sees the poses the fp32 engine was given (rounded to float)
        poses32 = poses.astype(np.float32).astype(np.float64)
        _, ut_ref = po.batch_control(cfg, MEANS, SIGMAS, MAP_BOUNDS, poses32, 0, min(threads, 64))
        du = float(np.abs(d_u0.cpu().numpy().astype(np.float64) - ut_ref[:, 0, :]).max())   # u0 = ut.col(0) (:310)
        dut = float(np.abs(d_ut.cpu().numpy().astype(np.float64) - ut_ref).max())
        worst[model] = (du, dut)
        eng.close()
    if os.environ.get("EEA_PRINT_WORST"):
        print("config 4 full size fp32 vs fp64 oracle (u0, ut)", worst)
    for model, (du, dut) in worst.items():
        assert du <= 1e-4 and dut <= 1e-4, worst


def _occupancy_seed2024(n=1024, block=32):
    """70 % free (0), 10 % occupied (100), 20 % unknown (-1) in 32 x 32 blocks"""
    rng = np.random.default_rng(2024)
    nb = n // block
    r = rng.random((nb, nb))
    cells = np.where(r < 0.7, 0, np.where(r < 0.8, 100, -1)).astype(np.int8)
    return np.ascontiguousarray(np.kron(cells, np.ones((block, block), dtype=np.int8)))


def test_config5_end_to_end_against_oracle():
    """1024 x 1024 occupancy -> entropy target -> phi_k (K = 30) -> control (T = 500) on the 102.4 m domain.
    phi_k at full size against eo_spatial_coeff (non-separated, ~1.9e9 cos calls: about half a minute), then
    4 agents x 2 consecutive control() calls stage by stage."""
    nx = ny = 1024
    res, lx, ly, K = 0.1, 102.4, 102.4, 30
    occ = _occupancy_seed2024(nx)
    lut = np.array([po.lib().eo_entropy(float(np.int8(np.uint8(b))) / 100.0) for b in range(256)])
    ent = lut[occ.reshape(-1).view(np.uint8)]
    phi_vals = ent / ent.sum()
    bounds = (0.0, lx, 0.0, ly)
    lim = np.array([1.0, 1.0, 2.0])
    Rinv = np.diag([1.0, 1.0, 2.0])
    B, horizon, dt = 4, 50.0, 0.1
    eng = capi.Engine(capi.make_config(capi.MODEL_OMNI, dt, horizon, res, 1.0, K, Rinv, -lim, lim))
    eng.set_target_occupancy(nx, ny, torch.as_tensor(occ).cuda(), lx, ly)
    assert not eng.config_domain(bounds)       # same extent: no rebuild, only map_pos is refreshed
    ors = []
    for b in range(B):
        o = po.ErgodicControl(po.MODEL_OMNI, dt, horizon, res, 1.0, K, Rinv, -lim, lim)
        if b == 0:
            o.set_target_grid(nx, ny, phi_vals, lx, ly)   # the expensive one
            phik_ref = o.phik
        ors.append(o)
    # 2^20 grid points: the reference formulation adds them one after the other (Armadillo sum(fk_mat, 1),
    # basis.cpp:132), so ITS rounding error is up to n eps / 2 * sum|terms| = 6e-11 on mode (0,0) (all terms
    # positive, sum 1); the kernel's (0,0) coefficient is exactly 1 (it is the normaliser).  Measured 1.4e-11
    # on that mode, <= 4e-15 on every other one.
    dphi = np.abs(eng.phik() - phik_ref)
    assert dphi[0] <= 1e-10 and dphi[1:].max() <= 1e-11, (dphi[0], dphi[1:].max())
    for o in ors[1:]:  # identical target: share the oracle's phi_k instead of recomputing it three times
        o.set_target_grid(1, 1, np.zeros(1), lx, ly)      # sets lx, ly (phi_k of a one-point zero grid) ...
        np.ctypeslib.as_array(po.lib().eo_control_phik(o.h), (K * K,))[:] = phik_ref  # ... then the real one
    T, K2 = eng.T, eng.K2
    assert T == 500
    rng = np.random.default_rng(5)
    poses = random_poses(rng, B, bounds)
    ut0 = rng.uniform(-0.5, 0.5, (B, T, 3))
    d_pose, d_ut = dev(poses), dev(ut0)
    d_u0 = torch.empty((B, 3), dtype=torch.float64, device="cuda")
    d_ck = torch.empty((B, K2), dtype=torch.float64, device="cuda")
    outs = {k: torch.empty((B, T, 3), dtype=torch.float64, device="cuda") for k in ("traj", "edx", "bdx", "rhot")}
    for b in range(B):
        ors[b].ut = ut0[b].T
    worst = {}

    def check(name, diff, bar, ctx):
        worst[name] = max(worst.get(name, 0.0), float(diff))
        assert diff <= bar, (name, diff, ctx)

    for call in range(2):
        eng.control_batch(B, d_pose, d_ut, d_u0, ck=d_ck, **outs)
        torch.cuda.synchronize()
        for b in range(B):
            u, st = ors[b].control(bounds, poses[b], None, stages=True)
            g = {k: v[b].cpu().numpy().T for k, v in outs.items()}
            check("traj_xy", np.abs(g["traj"][:2] - st["traj"][:2]).max(), TOL, (call, b))
            check("traj_th", np.abs(angle_diff(g["traj"][2], st["traj"][2])).max(), TOL, (call, b))
            check("ck", np.abs(d_ck[b].cpu().numpy() - st["ck"]).max(), TOL_CK, (call, b))
            for k in ("edx", "bdx", "rhot"):
                check(k, np.abs(g[k] - st[k]).max(), TOL * max(1.0, np.abs(st[k]).max()), (call, b))
            check("ut", np.abs(d_ut[b].cpu().numpy().T - st["ut"]).max(), TOL, (call, b))
            check("u0", np.abs(d_u0[b].cpu().numpy() - u).max(), TOL, (call, b))
            ors[b].ut = d_ut[b].cpu().numpy().T
    if os.environ.get("EEA_PRINT_WORST"):
        print("config 5 end to end: phi_k mode (0,0) %.2e, other modes %.2e;" % (dphi[0], dphi[1:].max()),
              {k: "%.2e" % v for k, v in worst.items()})
    eng.close()


def test_prepared_batch_is_the_same_call():
    """Engine.prepared_batch (bench.py's per-pass call: the eea_batch_io built once) issues exactly eea_control_batch:
    bitwise the same controls, first twists and c_k as control_batch on the same inputs, over consecutive passes, with
    replay memory, a shared c_k and a caller stream."""
    rng = np.random.default_rng(7)
    B, K, n_mem = 9, 10, 12
    eng, _ = make_pair("omni", K, 20.0, n_oracles=0)
    T, K2 = eng.T, eng.K2
    poses = random_poses(rng, B)
    ut0 = rng.uniform(-0.5, 0.5, (B, T, 3))
    mem = random_poses(rng, B * n_mem).reshape(B, n_mem, 3)
    cbar = rng.uniform(-0.1, 0.1, K2)
    stream = torch.cuda.Stream()
    res = []
    for prepared in (False, True):
        d_pose, d_ut, d_mem, d_cbar = dev(poses), dev(ut0), dev(mem), dev(cbar)
        d_nmem = torch.full((B,), n_mem, dtype=torch.int32, device="cuda")
        d_u0 = torch.empty((B, 3), dtype=torch.float64, device="cuda")
        d_ck = torch.empty((B, K2), dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        kw = dict(mem_cols=d_mem, n_mem=d_nmem, mem_stride=n_mem, ck=d_ck, ck_shared=d_cbar, stream=stream.cuda_stream)
        call = eng.prepared_batch(B, d_pose, d_ut, d_u0, **kw) if prepared else None
        for _ in range(3):
            if prepared:
                call()
            else:
                eng.control_batch(B, d_pose, d_ut, d_u0, **kw)
        torch.cuda.synchronize()
        res.append([t.cpu().numpy().copy() for t in (d_ut, d_u0, d_ck)])
    eng.close()
    for a, b in zip(*res):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("extra", [["--no-exchange"], []])
def test_bench_self_launches_its_ranks(extra, tmp_path):
    """`python bench.py --gpus 2` exactly as the driver would type it for N > 1 without a launcher: the
    parent makes no GPU call and starts the two ranks itself; on this 1-GPU box the ranks share the device
    (gloo rendezvous; the exchange steps through the C ABI and the RCCL test double).  One JSON line, n_gpus = 2, rc 0."""
    env = dict(os.environ, EEA_BENCH_DETAIL_DIR=str(tmp_path))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--passes-per-step", "3", "--exchange-passes-per-step", "3", "--agents", "256"] + extra
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-3000:]
    # the driver's line: small by construction (round 5's 22.5 KB line went unparsed), every leg's record in the detail file
    assert len(lines[0]) <= 4096 and r.stdout.rstrip("\n").splitlines()[-1] == lines[0]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["value"] > 0
    # the ranks' own times and values are in the line; value = all ranks' work / the max of the times
    pr = line["per_rank"]
    assert len(pr["values"]) == 2 and len(pr["timed_region_s"]) == 2
    assert line["value"] == pytest.approx(2 * 256 * 3 * 2 / max(pr["timed_region_s"]), rel=1e-3)
    with open(tmp_path / "bench_detail.json") as f:
        rec = json.load(f)
    assert rec["value"] == pytest.approx(line["value"], rel=1e-5)
    assert rec["config"]["optimisations_per_step"] == 2 * 256 * 3
    if extra:
        assert "exchange" not in rec and "exchange" not in line
    else:
        # the collective library itself reports both ranks (ncclCommCount through eea_comm_library_nranks)
        assert line["exchange"]["rccl_nranks"] == 2 and line["exchange"]["world"] == 2 and line["exchange"]["timeouts"] == 0
        assert rec["exchange"]["consensus_allreduce"]["value"] > 0
        assert rec["exchange"]["allgather_ck"]["value"] > 0
        # the ranks share the GPU, so the exchange steps go through the C ABI and the test double of
        # RCCL (collective kernels that meet on the device) -- not through gloo staged over the host -- and, with a collective
        # kernel in the exchange, no group waits INSIDE its control kernels: every consuming launch sits behind a one-wavefront
        # gate (the gated exchange), lag >= 2: no agent and no gate times out
        assert "test double" in rec["exchange"]["backend"], rec["exchange"]["backend"]
        ca = rec["exchange"]["consensus_allreduce"]
        assert "gated" in ca["consuming_groups"] and min(int(k) for k in ca["by_lag"]) >= 2
        assert all(v["agents_timed_out"] == 0 for v in ca["by_lag"].values()), ca["by_lag"]
        assert "test double" in rec["grid_tile"]["collective"]
        # BASELINE config 5 shard: the 1024 rows tiled over the two ranks, one all-reduce, phi_k installed on the device
        gt = rec["grid_tile"]
        assert gt["rows_per_rank"] == 512 and abs(gt["phik_00_check"] - 1.0) < 1e-12 and gt["us_per_rebuild_back_to_back"] > 0


@pytest.mark.parametrize("gpus", [1, 2])
def test_bench_headline_survives_a_stuck_exchange(gpus):
    """The exchange legs run last and under a watchdog: if they do not finish in time (here: a limit no run can meet;
    on a node whose collective library never returns it would be the 300 s default), rank 0 still prints the ONE line
    with the headline value and the reason under "exchange" -- and the processes exit NON-zero, so that the launcher
    and the driver see that a leg deadlocked (a process that gave up on a collective must not look like a clean run)."""
    env = dict(os.environ, EEA_BENCH_EXCHANGE_TIMEOUT="0.001")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "2", "--warmup", "1",
           "--passes-per-step", "3", "--exchange-passes-per-step", "3", "--agents", "256", "--cpu-seconds", "0",
           "--no-latency", "--no-phik"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode != 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-3000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == gpus and rec["value"] > 0 and "roofline" in rec
    assert "did not finish" in rec["exchange"]["error"]

