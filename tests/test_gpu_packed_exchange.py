"""Round 6: the several-agents-per-wavefront kernel (csrc/control_pack_impl.hpp) emits the agents' sum
records (eea_batch_io::d_ck_rec, + d_rec_ready for the device-bound exchange) and waits for the flag of the shared c_k it
consumes (d_ck_flag), so decentralised consensus (README ref. [2]; ergodic_control.hpp:418-436 with c_bar in place of c_k)
keeps packing at short horizons.  Checked per group size L = 8 / 16 / 32:
  * the records against the d_ck of the same call (bitwise), the count / pad elements, rejected agents (all-zero record, not
    counted), skipped agents (nothing written), ragged batches;
  * a two-pass consensus against the oracle's switch (eo_control_set_shared_ck) fed with the mean of the ORACLES' own c_k;
  * the stream-ordered and the device-bound consensus pass, as bench.py / AgentBatch issue them, bitwise the synchronised
    call-by-call sequence on the same kernel and <= 1e-9 the sequence of the wavefront-per-agent kernel."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from ergodic_exploration_amd import capi
from tests.gpu_util import MAP_BOUNDS, make_pair, random_poses
from tests.test_gpu_control_parity import TOL, TOL_CK, dev

pytestmark = pytest.mark.gpu

SHAPES = [(8, 20, "simple_cart"), (8, 5, "omni"), (16, 50, "omni"), (16, 33, "simple_cart"), (32, 50, "omni"), (32, 128, "simple_cart")]


@pytest.fixture
def lanes():
    def _set(v):
        capi.set_option(capi.OPT_AGENT_LANES, v)
    yield _set
    capi.set_option(capi.OPT_AGENT_LANES, 0)


def _inputs(rng, model, B, T):
    poses = random_poses(rng, B)
    ut0 = rng.uniform(-0.5, 0.5, (B, T, 3))
    if model == "simple_cart":
        ut0[:, :, 1] = 0.0
    return poses, ut0


@pytest.mark.parametrize("L,steps,model", SHAPES)
@pytest.mark.parametrize("bound", [False, True], ids=["plain", "ready-marks"])
def test_packed_sum_records(lanes, L, steps, model, bound):
    lanes(L)
    A = 64 // L
    B = 5 * A + 3   # ragged: the last wavefront is partly empty
    K = 5 if steps == 5 else 10
    eng, _ = make_pair(model, K, steps * 0.1, n_oracles=0)
    assert eng.agent_lanes(B) == L
    T, K2, RL = eng.T, eng.K2, eng.ck_record_len
    rng = np.random.default_rng(1000 + 7 * L + steps)
    poses, ut0 = _inputs(rng, model, B, T)
    bad = ()
    if model == "simple_cart":   # SimpleCart::operator() rejects these agents (cart.hpp:167-170)
        bad = (1, A, 2 * A + 1, B - 1)
        for b in bad:
            ut0[b, min(3, T - 1), 1] = 0.2
    skip = np.zeros(B, dtype=np.int32)
    skipped = (0, A + 1, 3 * A)
    skip[list(skipped)] = 1
    d_ut, d_u0 = dev(ut0), torch.empty((B, 3), dtype=torch.float64, device="cuda")
    d_ck = torch.full((B, K2), -3.0, dtype=torch.float64, device="cuda")
    d_rec = torch.full((B, RL), -5.0, dtype=torch.float64, device="cuda")
    d_ready = torch.zeros((B,), dtype=torch.int32, device="cuda")
    d_status = torch.full((B,), -1, dtype=torch.int32, device="cuda")
    kw = dict(rec_ready=d_ready, rec_seq=41) if bound else {}
    eng.control_batch(B, dev(poses), d_ut, d_u0, ck=d_ck, ck_rec=d_rec, status=d_status, skip=torch.as_tensor(skip).cuda(), **kw)
    torch.cuda.synchronize()
    rec, ck, st, ready = d_rec.cpu().numpy(), d_ck.cpu().numpy(), d_status.cpu().numpy(), d_ready.cpu().numpy()
    for b in range(B):
        if b in skipped:   # nothing of a skipped agent is read or written (also not its c_k)
            assert (rec[b] == -5.0).all() and (ck[b] == -3.0).all() and st[b] == -1 and ready[b] == 0
        elif b in bad:
            assert st[b] == capi.ERR_INVALID_TWIST and (rec[b] == 0.0).all() and (ck[b] == -3.0).all()
            assert ready[b] == (41 if bound else 0)
        else:
            assert st[b] == 0 and np.array_equal(rec[b, :K2], ck[b]) and rec[b, K2] == 1.0 and (rec[b, K2 + 1:] == 0.0).all()
            assert ready[b] == (41 if bound else 0)
    # the record sum of the launch = the good agents' c_k, counted
    live = [b for b in range(B) if b not in skipped]
    d_sum = torch.empty((RL,), dtype=torch.float64, device="cuda")
    idx = torch.as_tensor(live).cuda()
    eng.ck_records_sum(len(live), d_rec[idx].contiguous(), d_sum)
    torch.cuda.synchronize()
    s = d_sum.cpu().numpy()
    good = [b for b in live if b not in bad]
    assert s[K2] == len(good)
    assert np.abs(s[:K2] - ck[good].sum(0)).max() <= 1e-12 * max(1.0, np.abs(ck[good].sum(0)).max())
    # the same call on the wavefront-per-agent kernel: the records agree to the c_k bar (other summation trees)
    lanes(64)
    d_rec64 = torch.full((B, RL), -5.0, dtype=torch.float64, device="cuda")
    eng.control_batch(B, dev(poses), dev(ut0), d_u0, ck_rec=d_rec64, skip=torch.as_tensor(skip).cuda())
    torch.cuda.synchronize()
    assert np.abs(d_rec64.cpu().numpy() - rec).max() <= TOL_CK
    eng.close()


@pytest.mark.parametrize("L,steps,model", SHAPES)
@pytest.mark.parametrize("bound", [False, True], ids=["plain", "ready-marks"])
def test_packed_wavefront_records(lanes, L, steps, model, bound):
    """eea_batch_io::rec_per_wavefront: one record per wavefront = its accepted agents' per-agent records added in agent order
    (bitwise), their number in the count element; rejected and skipped agents stay out; a wavefront whose agents are all skipped
    writes nothing; one ready mark per record; controls and warm start are those of the per-agent call"""
    lanes(L)
    A = 64 // L
    B = 6 * A + (A + 1) // 2   # ragged: the last wavefront is partly empty
    K = 5 if steps == 5 else 10
    eng, _ = make_pair(model, K, steps * 0.1, n_oracles=0)
    assert eng.agent_lanes(B) == L
    W = eng.record_count(B)
    assert W == (B + A - 1) // A == 7
    T, K2, RL = eng.T, eng.K2, eng.ck_record_len
    rng = np.random.default_rng(4000 + 7 * L + steps)
    poses, ut0 = _inputs(rng, model, B, T)
    bad = ()
    if model == "simple_cart":
        bad = (1, A, 2 * A + 1, B - 1)
        for b in bad:
            ut0[b, min(3, T - 1), 1] = 0.2
    skip = np.zeros(B, dtype=np.int32)
    skipped = (0, A + 1, 3 * A) + tuple(range(4 * A, 5 * A))   # wavefront 4 is left out altogether
    skip[list(skipped)] = 1
    d_skip = torch.as_tensor(skip).cuda()
    out = {}
    for wave in (False, True):
        n = W if wave else B
        d_ut, d_u0 = dev(ut0), torch.full((B, 3), 9.0, dtype=torch.float64, device="cuda")
        d_rec = torch.full((n, RL), -5.0, dtype=torch.float64, device="cuda")
        d_ready = torch.zeros((n,), dtype=torch.int32, device="cuda")
        d_status = torch.full((B,), -1, dtype=torch.int32, device="cuda")
        kw = dict(rec_ready=d_ready, rec_seq=41) if bound else {}
        eng.control_batch(B, dev(poses), d_ut, d_u0, ck_rec=d_rec, rec_per_wavefront=wave, status=d_status, skip=d_skip, **kw)
        torch.cuda.synchronize()
        out[wave] = (d_ut, d_u0, d_rec.cpu().numpy(), d_ready.cpu().numpy(), d_status)
    assert torch.equal(out[False][0], out[True][0]) and torch.equal(out[False][1], out[True][1])
    assert torch.equal(out[False][4], out[True][4])
    arec, wrec, wready = out[False][2], out[True][2], out[True][3]
    n_good = 0
    for w in range(W):
        agents = [b for b in range(w * A, min(B, (w + 1) * A)) if b not in skipped]
        if not agents:
            assert (wrec[w] == -5.0).all() and wready[w] == 0
            continue
        good = [b for b in agents if b not in bad]
        want = np.zeros(K2)
        for b in good:            # agent order, one addition per agent: the kernel's order
            want = want + arec[b, :K2]
        assert np.array_equal(wrec[w, :K2], want), w
        assert wrec[w, K2] == len(good) and (wrec[w, K2 + 1:] == 0.0).all()
        assert wready[w] == (41 if bound else 0)
        n_good += len(good)
    # through the record sum: the touched wavefront records against the per-agent records of the same agents
    touched = [w for w in range(W) if (wrec[w] != -5.0).any()]
    d_sum = torch.empty((RL,), dtype=torch.float64, device="cuda")
    eng.ck_records_sum(len(touched), dev(wrec[touched]), d_sum)
    torch.cuda.synchronize()
    s = d_sum.cpu().numpy()
    live = [b for b in range(B) if b not in skipped and b not in bad]
    assert s[K2] == n_good == len(live)
    assert np.abs(s[:K2] - arec[live, :K2].sum(0)).max() <= 1e-12 * max(1.0, np.abs(arec[live, :K2].sum(0)).max())
    # kernels of one agent per wavefront write per-agent records either way
    lanes(64)
    assert eng.record_count(B) == B
    d_rec64 = torch.full((B, RL), -5.0, dtype=torch.float64, device="cuda")
    eng.control_batch(B, dev(poses), dev(ut0), torch.empty((B, 3), dtype=torch.float64, device="cuda"), ck_rec=d_rec64,
                      rec_per_wavefront=True, skip=d_skip)
    torch.cuda.synchronize()
    r64 = d_rec64.cpu().numpy()
    for b in range(B):
        if b in skipped:
            assert (r64[b] == -5.0).all()
        else:
            assert np.abs(r64[b] - arec[b]).max() <= TOL_CK
    eng.close()


@pytest.mark.parametrize("L,steps,model", SHAPES)
def test_packed_consensus_through_records_against_the_oracle(lanes, L, steps, model):
    """pass 1 leaves the records, eea_ck_records_sum adds them, pass 2 consumes the sum record (ck_shared_parts = 1); the
    oracles run the same two calls with c_bar = the mean of THEIR c_k (eo_control_set_shared_ck)"""
    lanes(L)
    A = 64 // L
    B = 2 * A + 1
    K = 5 if steps == 5 else 10
    eng, ors = make_pair(model, K, steps * 0.1, n_oracles=B)
    T, K2, RL = eng.T, eng.K2, eng.ck_record_len
    rng = np.random.default_rng(2000 + 7 * L + steps)
    poses, ut0 = _inputs(rng, model, B, T)
    d_pose, d_ut, d_u0 = dev(poses), dev(ut0), torch.empty((B, 3), dtype=torch.float64, device="cuda")
    d_rec = torch.zeros((B, RL), dtype=torch.float64, device="cuda")
    d_sum = torch.zeros((RL,), dtype=torch.float64, device="cuda")
    cks = []
    for b in range(B):
        ors[b].ut = ut0[b].T
        _, st = ors[b].control(MAP_BOUNDS, poses[b], None, stages=True)
        cks.append(st["ck"])
    cbar = np.mean(cks, axis=0)
    eng.control_batch(B, d_pose, d_ut, d_u0, ck_rec=d_rec)
    eng.ck_records_sum(B, d_rec, d_sum)
    eng.control_batch(B, d_pose, d_ut, d_u0, ck_rec=d_rec, ck_shared=d_sum, ck_shared_parts=1)
    torch.cuda.synchronize()
    s = d_sum.cpu().numpy()
    assert s[K2] == B and np.abs(s[:K2] / B - cbar).max() <= TOL_CK
    ut, u0 = d_ut.cpu().numpy(), d_u0.cpu().numpy()
    for b in range(B):
        ors[b].set_shared_ck(cbar)
        u, st = ors[b].control(MAP_BOUNDS, poses[b], None, stages=True)
        bar = TOL * max(1.0, np.abs(st["rhot"]).max())
        assert np.abs(ut[b].T - st["ut"]).max() <= bar and np.abs(u0[b] - u).max() <= bar
    eng.close()


def _record_slices(eng, gb, wave):
    """where each agent group's records go: per agent (its agents' rows) or per wavefront (eea_batch_record_count rows per group)"""
    if not wave:
        return [slice(gb[g], gb[g + 1]) for g in range(len(gb) - 1)], gb[-1]
    counts = [eng.record_count(gb[g + 1] - gb[g]) for g in range(len(gb) - 1)]
    offs = np.concatenate([[0], np.cumsum(counts)]).astype(int)
    return [slice(offs[g], offs[g + 1]) for g in range(len(counts))], int(offs[-1])


def _reference_sequence(eng, B, gb, d_pose, ut0, lag, passes, RL, wave=False):
    ut, u0 = dev(ut0), torch.empty((B, 3), dtype=torch.float64, device="cuda")
    rsl, n_rec = _record_slices(eng, gb, wave)
    arec = torch.zeros((n_rec, RL), dtype=torch.float64, device="cuda")
    sums = [torch.zeros((RL,), dtype=torch.float64, device="cuda") for _ in range(passes)]
    for i in range(passes):
        src = i - lag
        for g in range(len(gb) - 1):
            sl = slice(gb[g], gb[g + 1])
            eng.control_batch(gb[g + 1] - gb[g], d_pose[sl], ut[sl], u0[sl], ck_rec=arec[rsl[g]], rec_per_wavefront=wave,
                              ck_shared=sums[src] if src >= 0 else None, ck_shared_parts=1 if src >= 0 else 0)
        torch.cuda.synchronize()
        eng.ck_records_sum(n_rec, arec, sums[i])
        torch.cuda.synchronize()
    return ut, u0, sums


@pytest.mark.parametrize("L,steps,model", [(8, 20, "simple_cart"), (16, 50, "omni"), (32, 50, "omni")])
@pytest.mark.parametrize("mode", ["stream-ordered", "stream-ordered-rccl", "device-bound"])
@pytest.mark.parametrize("lag", [1, 2])
@pytest.mark.parametrize("wave", [False, True], ids=["agent-records", "wavefront-records"])
def test_packed_consensus_pass(lanes, L, steps, model, mode, lag, wave):
    """(wavefront-records: eea_batch_io::rec_per_wavefront -- one record and one ready mark per wavefront, the record sum and the
    exchange take eea_batch_record_count records per group instead of one per agent.)
    The consensus pass as bench.py issues it -- stream-ordered (eea_comm_records_exchange_async + eea_comm_wait; local and
    through a real one-rank RCCL communicator) and device-bound (ready marks out, flag wait in the kernel) -- on the packed
    kernel: bitwise the synchronised call-by-call sequence, no time-out; and <= 1e-9 the wavefront-per-agent kernel's"""
    lanes(L)
    A = 64 // L
    B, G, NB, passes = 37 * A + 5, 2, 6, 7
    eng, _ = make_pair(model, 10, steps * 0.1, n_oracles=0)
    assert eng.agent_lanes(B // 2) == L
    T, K2, RL = eng.T, eng.K2, eng.ck_record_len
    rng = np.random.default_rng(3000 + L + lag)
    poses, ut0 = _inputs(rng, model, B, T)
    ut0 *= 0.6
    d_pose = dev(poses)
    gb = [0, 17 * A + 2, B]
    ut_a, u0_a, sums_a = _reference_sequence(eng, B, gb, d_pose, ut0, lag, passes, RL, wave=wave)
    rsl, n_rec = _record_slices(eng, gb, wave)
    assert n_rec == ((gb[1] + A - 1) // A + (B - gb[1] + A - 1) // A if wave else B)
    comm = capi.Comm(0, 1, 0, capi.comm_unique_id() if mode.endswith("rccl") else None)
    streams = [torch.cuda.Stream() for _ in range(G)]
    ut_b, u0_b = dev(ut0), torch.empty((B, 3), dtype=torch.float64, device="cuda")
    arecs = [torch.zeros((n_rec, RL), dtype=torch.float64, device="cuda") for _ in range(NB)]
    sums_b = [torch.zeros((RL,), dtype=torch.float64, device="cuda") for _ in range(NB)]
    ready = torch.zeros((n_rec,), dtype=torch.int32, device="cuda")
    flag = torch.zeros((1,), dtype=torch.int32, device="cuda")
    status = torch.full((B,), -1, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    for i in range(passes):
        seq = i + 1
        slot, src = i % NB, (i - lag) % NB if i >= lag else None
        for g in range(G):
            sl = slice(gb[g], gb[g + 1])
            kw = dict(ck_rec=arecs[slot][rsl[g]], rec_per_wavefront=wave, status=status[sl],
                      ck_shared=None if src is None else sums_b[src],
                      ck_shared_parts=0 if src is None else 1, stream=streams[g].cuda_stream)
            if mode == "device-bound":
                kw.update(rec_ready=ready[rsl[g]], rec_seq=seq, ck_flag=None if src is None else flag, ck_flag_seq=seq - lag)
            elif src is not None:
                comm.wait(src, streams[g].cuda_stream)
            eng.control_batch(gb[g + 1] - gb[g], d_pose[sl], ut_b[sl], u0_b[sl], **kw)
        if mode == "device-bound":
            comm.records_exchange_bound(eng, n_rec, arecs[slot], ready, seq, sums_b[slot], flag, slot)
        else:
            comm.records_exchange_async(eng, n_rec, arecs[slot], sums_b[slot], [st.cuda_stream for st in streams], slot)
    torch.cuda.synchronize()
    assert (status.cpu().numpy() == 0).all()
    assert torch.equal(ut_a, ut_b) and torch.equal(u0_a, u0_b)
    assert torch.equal(sums_a[passes - 1], sums_b[(passes - 1) % NB])
    assert float(sums_a[passes - 1][K2]) == B
    # the same sequence on the wavefront-per-agent kernel
    lanes(64)
    ut_c, u0_c, sums_c = _reference_sequence(eng, B, gb, d_pose, ut0, lag, passes, RL)
    scale = max(1.0, float(ut_c.abs().max()))
    assert float((ut_c - ut_a).abs().max()) <= 1e-8 * scale and float((u0_c - u0_a).abs().max()) <= 1e-8 * scale
    comm.close()
    eng.close()


def test_packed_flag_wait_times_out_instead_of_hanging(lanes):
    """a flag that never arrives: every agent of the packed wavefronts goes on with its own c_k and reports EEA_ERR_TIMEOUT"""
    lanes(16)
    eng, _ = make_pair("omni", 10, 5.0, n_oracles=0)
    B, T, RL = 21, eng.T, eng.ck_record_len
    rng = np.random.default_rng(5)
    poses, ut0 = _inputs(rng, "omni", B, T)
    d_u0 = torch.empty((B, 3), dtype=torch.float64, device="cuda")
    status = torch.zeros((B,), dtype=torch.int32, device="cuda")
    flag = torch.zeros((1,), dtype=torch.int32, device="cuda")
    rec = torch.zeros((RL,), dtype=torch.float64, device="cuda")
    rec[0] = 123.0
    rec[eng.K2] = 1.0
    d_ut = dev(ut0)
    eng.control_batch(B, dev(poses), d_ut, d_u0, status=status, ck_shared=rec, ck_shared_parts=1, ck_flag=flag, ck_flag_seq=5)
    torch.cuda.synchronize()
    assert (status.cpu().numpy() == capi.ERR_TIMEOUT).all()
    d_ut2 = dev(ut0)
    eng.control_batch(B, dev(poses), d_ut2, d_u0)   # own c_k
    torch.cuda.synchronize()
    assert torch.equal(d_ut, d_ut2)
    eng.close()
