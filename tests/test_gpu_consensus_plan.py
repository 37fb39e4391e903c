"""eea_consensus_plan (ABI 6): the consensus passes of a rank as ONE replayable device graph -- the
stream-ordered protocol (per pass: the groups' control launches with records out and the sum record of pass i - lag in, the
record sum, the all-reduce over the ranks) captured once and replayed with one runtime call per `passes_per_launch` passes.
Bar: BITWISE the same controls, warm starts and sum records as the same passes issued call by call with explicit
synchronisation (same kernels, same summation tree), over several launches (the protocol continues across launches), for the
wavefront-per-agent kernel, the packed kernel (short horizons) and the workgroup kernel (K = 30); with the local communicator and
with a real one-rank RCCL communicator (ncclAllReduce captured into the graph).
Reference semantics: decentralised ergodic control shares c_k (README.md:225-227); ergodic_control.hpp:418-436 with c_bar."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from ergodic_exploration_amd import capi
from tests.gpu_util import make_pair, random_poses
from tests.test_gpu_control_parity import dev

pytestmark = pytest.mark.gpu


def _reference(eng, B, gb, d_pose, ut0, lag, passes, RL, status=None):
    """the passes one call at a time: pass i consumes the sum record of pass i - lag (an empty record before there is one).
    Records as the plan asks for them: one per wavefront where agents share one (eea_batch_io::rec_per_wavefront), else per agent"""
    ut, u0 = dev(ut0), torch.empty((B, 3), dtype=torch.float64, device="cuda")
    counts = [eng.record_count(gb[g + 1] - gb[g]) for g in range(len(gb) - 1)]
    roff = np.concatenate([[0], np.cumsum(counts)]).astype(int)
    n_rec = int(roff[-1])
    arec = torch.zeros((n_rec, RL), dtype=torch.float64, device="cuda")
    sums = [torch.zeros((RL,), dtype=torch.float64, device="cuda") for _ in range(passes)]
    empty = torch.zeros((RL,), dtype=torch.float64, device="cuda")
    for i in range(passes):
        src = i - lag
        for g in range(len(gb) - 1):
            sl = slice(gb[g], gb[g + 1])
            eng.control_batch(gb[g + 1] - gb[g], d_pose[sl], ut[sl], u0[sl], ck_rec=arec[roff[g]:roff[g + 1]], rec_per_wavefront=True,
                              status=None if status is None else status[sl],
                              ck_shared=sums[src] if src >= 0 else empty, ck_shared_parts=1)
        torch.cuda.synchronize()
        eng.ck_records_sum(n_rec, arec, sums[i])
        torch.cuda.synchronize()
    return ut, u0, sums


@pytest.mark.parametrize("model,K,horizon,B,lanes", [
    ("simple_cart", 10, 20.0, 300, 0),     # the metric shape: wavefront per agent
    ("omni", 10, 5.0, 8 * 37 + 5, 16),     # yaml T = 50 on the packed kernel, 16 lanes per agent
    ("simple_cart", 10, 2.0, 8 * 41 + 3, 8),   # configs[1] packed, 8 lanes per agent
    ("omni", 30, 6.0, 90, 0),              # workgroup-per-agent kernel
])
@pytest.mark.parametrize("lag,rccl", [(2, False), (2, True), (1, False), (3, True)])
def test_plan_is_bitwise_the_call_by_call_sequence(model, K, horizon, B, lanes, lag, rccl):
    capi.set_option(capi.OPT_AGENT_LANES, lanes)
    try:
        eng, _ = make_pair(model, K, horizon, n_oracles=0)
        T, K2, RL = eng.T, eng.K2, eng.ck_record_len
        rng = np.random.default_rng(50 + K + lag)
        d_pose = dev(random_poses(rng, B))
        ut0 = rng.uniform(-0.3, 0.3, (B, T, 3))
        if model == "simple_cart":
            ut0[:, :, 1] = 0.0
        gb = [0, B // 2 - 7, B]
        if lanes:
            assert eng.agent_lanes(gb[1]) == lanes
        slots = lag + 2
        per = 2 * slots                      # passes per launch
        launches = 3
        ut_a, u0_a, sums_a = _reference(eng, B, gb, d_pose, ut0, lag, per * launches, RL)
        comm = capi.Comm(0, 1, 0, capi.comm_unique_id() if rccl else None)
        ut_b, u0_b = dev(ut0), torch.empty((B, 3), dtype=torch.float64, device="cuda")
        status = torch.full((B,), -1, dtype=torch.int32, device="cuda")
        groups = [dict(B=gb[g + 1] - gb[g], pose=d_pose[gb[g]:gb[g + 1]], ut=ut_b[gb[g]:gb[g + 1]], u0=u0_b[gb[g]:gb[g + 1]],
                       status=status[gb[g]:gb[g + 1]]) for g in range(2)]
        plan = capi.ConsensusPlan(eng, comm, groups, lag=lag, passes_per_launch=per - 1)   # rounded up to a multiple of the slots
        assert plan.passes_per_launch == per
        st = torch.cuda.Stream()
        for n in range(launches):
            plan.launch(st.cuda_stream)
            if n == 0:   # the protocol continues across launches: the intermediate state is the reference's, too
                torch.cuda.synchronize()
                assert np.array_equal(plan.last_sum(eng), sums_a[per - 1].cpu().numpy())
        torch.cuda.synchronize()
        assert (status.cpu().numpy() == 0).all()
        assert torch.equal(ut_a, ut_b) and torch.equal(u0_a, u0_b)
        last = plan.last_sum(eng)
        assert np.array_equal(last, sums_a[per * launches - 1].cpu().numpy()) and last[K2] == B
        plan.close()
        comm.close()
        eng.close()
    finally:
        capi.set_option(capi.OPT_AGENT_LANES, 0)


def test_plan_with_replay_memory_rejected_agents_and_new_poses():
    """the caller's buffers are read at replay time: poses rewritten between launches are used; ragged replay memory; agents
    SimpleCart rejects (cart.hpp:167-170) stay out of the sum (count) and report EEA_ERR_INVALID_TWIST in the caller's status"""
    eng, _ = make_pair("simple_cart", 10, 20.0, n_oracles=0)
    B, T, K2, RL, lag = 140, eng.T, eng.K2, eng.ck_record_len, 2
    rng = np.random.default_rng(77)
    poses = [random_poses(rng, B) for _ in range(2)]
    ut0 = rng.uniform(-0.3, 0.3, (B, T, 3))
    ut0[:, :, 1] = 0.0
    bad = (3, 70, 139)
    for b in bad:
        ut0[b, 5, 1] = 0.3
    n_mem = rng.integers(0, 9, B).astype(np.int32)
    mem = random_poses(rng, B * 8).reshape(B, 8, 3)
    d_mem, d_nmem = dev(mem), torch.as_tensor(n_mem).cuda()
    gb = [0, 64, B]
    per = 4
    # reference: call by call, the pose buffer rewritten after the first `per` passes
    ut_a, u0_a = dev(ut0), torch.zeros((B, 3), dtype=torch.float64, device="cuda")   # (a rejected agent's u0 is never written)
    arec = torch.zeros((B, RL), dtype=torch.float64, device="cuda")
    sums = [torch.zeros((RL,), dtype=torch.float64, device="cuda") for _ in range(2 * per)]
    empty = torch.zeros((RL,), dtype=torch.float64, device="cuda")
    st_a = torch.full((B,), -1, dtype=torch.int32, device="cuda")
    for i in range(2 * per):
        d_pose = dev(poses[i // per])
        for g in range(2):
            sl = slice(gb[g], gb[g + 1])
            eng.control_batch(gb[g + 1] - gb[g], d_pose[sl], ut_a[sl], u0_a[sl], mem_cols=d_mem[sl], n_mem=d_nmem[sl], mem_stride=8,
                              ck_rec=arec[sl], status=st_a[sl], ck_shared=sums[i - lag] if i >= lag else empty, ck_shared_parts=1)
        torch.cuda.synchronize()
        eng.ck_records_sum(B, arec, sums[i])
        torch.cuda.synchronize()
    comm = capi.Comm(0, 1, 0, None)
    d_pose = dev(poses[0])
    ut_b, u0_b = dev(ut0), torch.zeros((B, 3), dtype=torch.float64, device="cuda")
    st_b = torch.full((B,), -1, dtype=torch.int32, device="cuda")
    groups = [dict(B=gb[g + 1] - gb[g], pose=d_pose[gb[g]:gb[g + 1]], ut=ut_b[gb[g]:gb[g + 1]], u0=u0_b[gb[g]:gb[g + 1]],
                   mem_cols=d_mem[gb[g]:gb[g + 1]], n_mem=d_nmem[gb[g]:gb[g + 1]], mem_stride=8, status=st_b[gb[g]:gb[g + 1]])
              for g in range(2)]
    plan = capi.ConsensusPlan(eng, comm, groups, lag=lag, passes_per_launch=per)
    plan.launch()
    torch.cuda.synchronize()
    d_pose.copy_(dev(poses[1]))      # same buffer, new contents
    torch.cuda.synchronize()
    plan.launch()
    torch.cuda.synchronize()
    assert torch.equal(st_a, st_b) and sorted(np.nonzero(st_b.cpu().numpy() == capi.ERR_INVALID_TWIST)[0]) == list(bad)
    assert torch.equal(ut_a, ut_b) and torch.equal(u0_a, u0_b)
    last = plan.last_sum(eng)
    assert np.array_equal(last, sums[-1].cpu().numpy()) and last[K2] == B - len(bad)
    plan.close()
    comm.close()
    eng.close()


def test_plan_argument_errors():
    eng, _ = make_pair("omni", 10, 5.0, n_oracles=0)
    comm = capi.Comm(0, 1, 0, None)
    B, T = 16, eng.T
    t = dict(B=B, pose=torch.zeros((B, 3), dtype=torch.float64, device="cuda"), ut=torch.zeros((B, T, 3), dtype=torch.float64, device="cuda"),
             u0=torch.zeros((B, 3), dtype=torch.float64, device="cuda"))
    for kw in (dict(lag=0), dict(lag=7), dict(passes_per_launch=0), dict(passes_per_launch=5000)):
        with pytest.raises(capi.EngineError) as ei:
            capi.ConsensusPlan(eng, comm, [t], **kw)
        assert ei.value.status == capi.ERR_INVALID_ARGUMENT
    with pytest.raises(capi.EngineError):
        capi.ConsensusPlan(eng, comm, [dict(t, B=0)])
    with pytest.raises(capi.EngineError):
        capi.ConsensusPlan(eng, comm, [dict(t, ut=None)])
    with pytest.raises(capi.EngineError):
        capi.ConsensusPlan(eng, comm, [t] * 9)
    plan = capi.ConsensusPlan(eng, comm, [t], lag=1, passes_per_launch=3)   # a single group is fine
    assert plan.passes_per_launch == 3
    plan.launch()
    torch.cuda.synchronize()
    plan.close()
    comm.close()
    eng.close()


def test_records_sum_with_a_caller_owned_workspace():
    """eea_ck_records_sum_ws: the same bits as eea_ck_records_sum (same tree), back to back into the same output"""
    import ctypes as C
    eng, _ = make_pair("omni", 10, 5.0, n_oracles=0)
    RL = eng.ck_record_len
    for B in (1, 33, 300, 4099):
        rng = np.random.default_rng(B)
        a = dev(rng.standard_normal((B, RL)) * 10.0 ** rng.integers(-3, 4, (B, 1)))
        ref = torch.empty((RL,), dtype=torch.float64, device="cuda")
        eng.ck_records_sum(B, a, ref)
        wb, tb = C.c_size_t(), C.c_size_t()
        capi.check(capi.lib().eea_ck_records_sum_ws_bytes(eng.h, B, C.byref(wb), C.byref(tb)))
        ws = torch.empty((wb.value,), dtype=torch.uint8, device="cuda")
        tk = torch.zeros((tb.value,), dtype=torch.uint8, device="cuda")
        out = torch.full((RL,), float("nan"), dtype=torch.float64, device="cuda")
        for _ in range(3):
            capi.check(capi.lib().eea_ck_records_sum_ws(eng.h, B, a.data_ptr(), out.data_ptr(), ws.data_ptr(), tk.data_ptr(), None))
        torch.cuda.synchronize()
        assert torch.equal(ref, out)
    eng.close()


@pytest.mark.parametrize("model,K,horizon,B,lanes", [
    ("simple_cart", 10, 20.0, 300, 0),
    ("omni", 10, 5.0, 8 * 37 + 5, 16),     # the packed kernel: ready marks out of several agents per wavefront
    ("omni", 30, 6.0, 90, 0),              # workgroup-per-agent kernel
])
@pytest.mark.parametrize("lag,rccl", [(2, False), (2, True), (1, True), (4, False)])
def test_gated_exchange_is_bitwise_the_call_by_call_sequence(model, K, horizon, B, lanes, lag, rccl):
    """The GATED exchange (eea_stream_wait_flag, ABI 6): per pass and group a one-wavefront gate on the group stream (returns
    once the flag of pass i - lag is published), the control launch (records + ready marks out, the gated record in, NO in-kernel
    flag wait) and ONE eea_comm_records_exchange_bound -- launches only.  Bitwise the synchronised sequence; no gate time-out."""
    capi.set_option(capi.OPT_AGENT_LANES, lanes)
    try:
        eng, _ = make_pair(model, K, horizon, n_oracles=0)
        T, K2, RL = eng.T, eng.K2, eng.ck_record_len
        rng = np.random.default_rng(90 + K + lag)
        d_pose = dev(random_poses(rng, B))
        ut0 = rng.uniform(-0.3, 0.3, (B, T, 3))
        if model == "simple_cart":
            ut0[:, :, 1] = 0.0
        gb = [0, B // 2 - 7, B]
        NB, passes = 6, 11
        ut_a, u0_a, sums_a = _reference(eng, B, gb, d_pose, ut0, lag, passes, RL)
        comm = capi.Comm(0, 1, 0, capi.comm_unique_id() if rccl else None)
        streams = [torch.cuda.Stream() for _ in range(2)]
        ut_b, u0_b = dev(ut0), torch.empty((B, 3), dtype=torch.float64, device="cuda")
        # (records as in _reference: one per wavefront where agents share one)
        counts = [eng.record_count(gb[g + 1] - gb[g]) for g in range(2)]
        roff = [0, counts[0], counts[0] + counts[1]]
        n_rec = roff[2]
        arecs = [torch.zeros((n_rec, RL), dtype=torch.float64, device="cuda") for _ in range(NB)]
        sums_b = [torch.zeros((RL,), dtype=torch.float64, device="cuda") for _ in range(NB)]
        ready = torch.zeros((n_rec,), dtype=torch.int32, device="cuda")
        flag = torch.zeros((1,), dtype=torch.int32, device="cuda")
        timeouts = torch.zeros((1,), dtype=torch.int32, device="cuda")
        status = torch.full((B,), -1, dtype=torch.int32, device="cuda")
        empty = torch.zeros((RL,), dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        for i in range(passes):
            seq = i + 1
            slot, src = i % NB, (i - lag) % NB if i >= lag else None
            for g in range(2):
                sl = slice(gb[g], gb[g + 1])
                if src is not None:
                    capi.stream_wait_flag(flag, seq - lag, timeouts, streams[g].cuda_stream)
                rs = slice(roff[g], roff[g + 1])
                eng.control_batch(gb[g + 1] - gb[g], d_pose[sl], ut_b[sl], u0_b[sl], ck_rec=arecs[slot][rs], rec_ready=ready[rs],
                                  rec_per_wavefront=True, rec_seq=seq, status=status[sl],
                                  ck_shared=empty if src is None else sums_b[src], ck_shared_parts=1, stream=streams[g].cuda_stream)
            comm.records_exchange_bound(eng, n_rec, arecs[slot], ready, seq, sums_b[slot], flag, slot)
        torch.cuda.synchronize()
        assert int(timeouts.item()) == 0 and (status.cpu().numpy() == 0).all()
        assert int(flag.item()) == passes
        assert torch.equal(ut_a, ut_b) and torch.equal(u0_a, u0_b)
        assert torch.equal(sums_a[passes - 1], sums_b[(passes - 1) % NB])
        comm.close()
        eng.close()
    finally:
        capi.set_option(capi.OPT_AGENT_LANES, 0)


def test_gate_times_out_instead_of_hanging():
    """a flag that never arrives: the gate gives up after about a second, counts it, and the stream goes on"""
    import time
    flag = torch.zeros((1,), dtype=torch.int32, device="cuda")
    timeouts = torch.zeros((1,), dtype=torch.int32, device="cuda")
    st = torch.cuda.Stream()
    t0 = time.perf_counter()
    capi.stream_wait_flag(flag, 5, timeouts, st.cuda_stream)
    capi.stream_wait_flag(flag, 0, timeouts, st.cuda_stream)    # satisfied at once
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert int(timeouts.item()) == 1 and 0.2 < dt < 10.0
    flag.fill_(7)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    capi.stream_wait_flag(flag, 5, timeouts, st.cuda_stream)
    capi.stream_wait_flag(flag, 0xfffffff0, timeouts, st.cuda_stream)   # sequence numbers wrap: 7 - 0xfffffff0 >= 0 (mod 2^32)
    torch.cuda.synchronize()
    assert int(timeouts.item()) == 1 and time.perf_counter() - t0 < 0.2
    with pytest.raises(capi.EngineError):
        capi.stream_wait_flag(None, 0)
