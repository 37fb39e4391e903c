"""Several agents per wavefront (csrc/control_pack_impl.hpp; round 5) against the CPU oracle.

Short horizons share a wavefront: an agent is a group of L = 8 / 16 / 32 lanes, a lane owns up to ceil(T / L) <= 4 steps.
Each group size is forced through EEA_OPT_AGENT_LANES (the engine's own choice depends on the batch size: the last tests)
and run through `run_batch_vs_oracle` -- stage by stage and as the instances without stage outputs that bench.py times --
over T = 2 .. 4 L, both models, replay memory 0 / 7 / 100 (ragged per agent), batch sizes that are not multiples of the
agents per wavefront.  Bars: the oracle bars of test_gpu_control_parity.py (c_k <= 1e-11, everything else <= 1e-9
max(1, |stage|)); bitwise equality with the wavefront-per-agent kernel is NOT required (other summation trees).
Reference: ergodic_control.hpp:224-311; the shipped operating point config/explore_omni.yaml:49-56 (K = 10, T = 50).
"""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from ergodic_exploration_amd import capi
from tests.gpu_util import MAP_BOUNDS, make_pair, random_poses
from tests.test_gpu_control_parity import TOL, TOL_CK, dev, run_batch_vs_oracle

pytestmark = pytest.mark.gpu


@pytest.fixture
def lanes():
    """sets EEA_OPT_AGENT_LANES for the test, restores the automatic choice afterwards"""
    def _set(v):
        capi.set_option(capi.OPT_AGENT_LANES, v)
    yield _set
    capi.set_option(capi.OPT_AGENT_LANES, 0)


def _check_lanes(model, K, steps, dt, L, B):
    eng, _ = make_pair(model, K, steps * dt, dt=dt, n_oracles=0)
    try:
        assert eng.T == steps
        assert eng.agent_lanes(B) == L, (eng.agent_lanes(B), L, steps)
    finally:
        eng.close()


# horizons per group size: around every steps-per-lane boundary (L, 2 L, 3 L, 4 L), the top-heavy horizons 3 L + 1 .. 3 L + L / 8 (cooperative tail gradient), partial second instruction group
# (T - j <= S L / 2), the smallest horizon the reference accepts (2) and the BASELINE shapes (5, 20, 50)
HORIZONS = {8: [2, 3, 4, 5, 7, 8, 9, 12, 13, 16, 17, 20, 23, 24, 25, 29, 32],
            16: [2, 5, 8, 9, 15, 16, 17, 20, 31, 32, 33, 40, 47, 48, 49, 50, 51, 63, 64],
            32: [2, 5, 16, 17, 20, 31, 32, 33, 50, 64, 65, 95, 96, 97, 98, 99, 100, 101, 127, 128]}
CASES = [(L, T) for L in (8, 16, 32) for T in HORIZONS[L]]


@pytest.mark.parametrize("L,steps", CASES)
def test_packed_stagewise_and_timed_instances(lanes, L, steps):
    lanes(L)
    A = 64 // L
    dt = 0.125   # horizon / dt exact
    B = 2 * A + 1 + (steps % A)   # never a multiple of the agents per wavefront
    _check_lanes("simple_cart", 10, steps, dt, L, B)
    run_batch_vs_oracle("simple_cart", 10, steps * dt, dt, B=B, n_mem=0, calls=2, seed=100 + steps)
    run_batch_vs_oracle("omni", 10, steps * dt, dt, B=B, n_mem=7, calls=2, seed=200 + steps, stages=False)
    run_batch_vs_oracle("omni", 5, steps * dt, dt, B=A + 1, n_mem=0, calls=2, seed=300 + steps, stages=False)


@pytest.mark.parametrize("L,steps", [(8, 5), (8, 20), (8, 32), (16, 20), (16, 50), (16, 64), (32, 50), (32, 100), (32, 128)])
@pytest.mark.parametrize("n_mem", [7, 100])
def test_packed_replay_memory(lanes, L, steps, n_mem):
    """replay-memory columns in rounds of L per agent: fewer than one round, many rounds, both models, K = 5 and 10"""
    lanes(L)
    A = 64 // L
    dt = 0.1 if steps in (5, 20, 50, 100) else 0.125
    run_batch_vs_oracle("omni", 10, steps * dt, dt, B=A + 3, n_mem=n_mem, calls=2, seed=7 * steps + n_mem)
    run_batch_vs_oracle("simple_cart", 5, steps * dt, dt, B=2 * A - 1, n_mem=n_mem, calls=2, seed=9 * steps + n_mem, stages=False)
    run_batch_vs_oracle("simple_cart", 10, steps * dt, dt, B=3 * A + 1, n_mem=n_mem, calls=2, seed=11 * steps + n_mem, stages=False)


@pytest.mark.parametrize("L,steps", [(8, 20), (16, 50), (32, 50)])
def test_packed_ragged_memory_counts(lanes, L, steps):
    """every agent of a wavefront its own number of memory columns (0 .. mem_stride), incl. agents with none"""
    lanes(L)
    A = 64 // L
    B = 2 * A + 1
    rng = np.random.default_rng(5 + L)
    eng, ors = make_pair("omni", 10, steps * 0.1, n_oracles=B)
    T, K2, stride = eng.T, eng.K2, 37
    assert eng.agent_lanes(B) == L
    poses = random_poses(rng, B)
    ut0 = rng.uniform(-0.5, 0.5, (B, T, 3))
    mem = random_poses(rng, B * stride).reshape(B, stride, 3)
    counts = rng.integers(0, stride + 1, B)
    counts[0], counts[1], counts[-1] = 0, stride, 1
    d_ut, d_u0 = dev(ut0), torch.empty((B, 3), dtype=torch.float64, device="cuda")
    d_ck = torch.empty((B, K2), dtype=torch.float64, device="cuda")
    eng.control_batch(B, dev(poses), d_ut, d_u0, mem_cols=dev(mem), n_mem=torch.as_tensor(counts, dtype=torch.int32).cuda(),
                      mem_stride=stride, ck=d_ck)
    torch.cuda.synchronize()
    ck, ut, u0 = d_ck.cpu().numpy(), d_ut.cpu().numpy(), d_u0.cpu().numpy()
    for b in range(B):
        ors[b].ut = ut0[b].T
        u, st = ors[b].control(MAP_BOUNDS, poses[b], mem[b, :counts[b]].T if counts[b] else None, stages=True)
        assert np.abs(ck[b] - st["ck"]).max() <= TOL_CK
        assert np.abs(ut[b].T - st["ut"]).max() <= TOL * max(1.0, np.abs(st["rhot"]).max())
        assert np.abs(u0[b] - u).max() <= TOL * max(1.0, np.abs(st["rhot"]).max())
    eng.close()


@pytest.mark.parametrize("L,steps,K", [(8, 5, 5), (8, 20, 10), (16, 20, 10), (16, 50, 10), (32, 50, 10), (32, 20, 5)])
def test_packed_steps_in_one_launch_equal_separate_calls(lanes, L, steps, K):
    """eea_control_batch_steps on the packed kernel: bitwise the separate launches (pose sequence and fixed pose)"""
    lanes(L)
    A = 64 // L
    B, n_steps, n_mem = 5 * A + 3, 5, 9
    for model in ("simple_cart", "omni"):
        eng, _ = make_pair(model, K, steps * 0.1, n_oracles=0)
        assert eng.agent_lanes(B) == L
        rng = np.random.default_rng(3 * steps + K)
        T = eng.T
        pose0 = random_poses(rng, B)
        seq = pose0[None] + np.cumsum(rng.normal(scale=0.02, size=(n_steps, B, 3)), axis=0)
        ut0 = rng.uniform(-0.4, 0.4, (B, T, 3))
        if model == "simple_cart":
            ut0[:, :, 1] = 0.0
        mem = random_poses(rng, B * n_mem).reshape(B, n_mem, 3)
        d_seq, d_mem = dev(seq), dev(mem)
        d_nmem = torch.full((B,), n_mem, dtype=torch.int32, device="cuda")
        kw = dict(mem_cols=d_mem, n_mem=d_nmem, mem_stride=n_mem)
        ut_a, ut_b = dev(ut0), dev(ut0)
        u0_a = torch.empty((n_steps, B, 3), dtype=torch.float64, device="cuda")
        u0_b = torch.full((n_steps, B, 3), float("nan"), dtype=torch.float64, device="cuda")
        for n in range(n_steps):
            eng.control_batch(B, d_seq[n], ut_a, u0_a[n], **kw)
        eng.control_batch(B, d_seq, ut_b, u0_b, n_steps=n_steps, pose_step_stride=B, u0_step_stride=B, **kw)
        torch.cuda.synchronize()
        assert torch.equal(ut_a, ut_b) and torch.equal(u0_a, u0_b)
        ut_c, ut_d = dev(ut0), dev(ut0)
        u0_c = torch.empty((B, 3), dtype=torch.float64, device="cuda")
        u0_d = torch.empty((B, 3), dtype=torch.float64, device="cuda")
        for n in range(n_steps):
            eng.control_batch(B, d_seq[0], ut_c, u0_c, **kw)
        eng.control_batch(B, d_seq[0], ut_d, u0_d, n_steps=n_steps, **kw)
        torch.cuda.synchronize()
        assert torch.equal(ut_c, ut_d) and torch.equal(u0_c, u0_d)
        eng.close()


@pytest.mark.parametrize("L,steps", [(8, 20), (16, 20), (32, 50)])
def test_packed_agents_do_not_leak_into_each_other(lanes, L, steps):
    """(a) SimpleCart::operator()'s throw (cart.hpp:167-170) is per agent: status 2, its buffers untouched, every other agent
    of the same wavefront bitwise what it is without the bad neighbour; (b) an agent whose pose is NaN / inf ends with NaN
    controls (std::clamp lets a NaN pass, ergodic_control.hpp:447-449) and leaves its wavefront neighbours bitwise alone."""
    lanes(L)
    A = 64 // L
    B = 3 * A
    eng, _ = make_pair("simple_cart", 10, steps * 0.1, n_oracles=0)
    assert eng.agent_lanes(B) == L
    T = eng.T
    rng = np.random.default_rng(77)
    poses = random_poses(rng, B)
    ut0 = rng.uniform(-0.5, 0.5, (B, T, 3))
    ut0[:, :, 1] = 0.0

    def run(p, u):
        d_ut, d_u0 = dev(u), torch.full((B, 3), -7.0, dtype=torch.float64, device="cuda")
        d_st = torch.full((B,), -1, dtype=torch.int32, device="cuda")
        eng.control_batch(B, dev(p), d_ut, d_u0, status=d_st)
        torch.cuda.synchronize()
        return d_ut.cpu().numpy(), d_u0.cpu().numpy(), d_st.cpu().numpy()

    ut_ref, u0_ref, st_ref = run(poses, ut0)
    assert (st_ref == 0).all()
    # (a) a lateral velocity in one agent of the first and one of the last wavefront
    bad = [1, B - 1]
    ut_bad = ut0.copy()
    ut_bad[bad[0], min(3, T - 1), 1] = 0.2
    ut_bad[bad[1], T - 1, 1] = -1e-3
    ut_a, u0_a, st_a = run(poses, ut_bad)
    for b in range(B):
        if b in bad:
            assert st_a[b] == capi.ERR_INVALID_TWIST
            assert np.array_equal(ut_a[b], ut_bad[b]) and (u0_a[b] == -7.0).all()
        else:
            assert st_a[b] == 0 and np.array_equal(ut_a[b], ut_ref[b]) and np.array_equal(u0_a[b], u0_ref[b])
    # (a') the same in a multi-step launch: rejected in step 0, out for the rest of the launch (its wavefront neighbours carry on)
    n_steps = 3
    d_ut_r, d_ut_b = dev(ut0), dev(ut_bad)
    d_u0_r = torch.full((n_steps, B, 3), -7.0, dtype=torch.float64, device="cuda")
    d_u0_b = torch.full((n_steps, B, 3), -7.0, dtype=torch.float64, device="cuda")
    d_st = torch.full((B,), -1, dtype=torch.int32, device="cuda")
    eng.control_batch(B, dev(poses), d_ut_r, d_u0_r, n_steps=n_steps, u0_step_stride=B)
    eng.control_batch(B, dev(poses), d_ut_b, d_u0_b, n_steps=n_steps, u0_step_stride=B, status=d_st)
    torch.cuda.synchronize()
    ut_r, ut_m, u0_r, u0_m, st_m = (t.cpu().numpy() for t in (d_ut_r, d_ut_b, d_u0_r, d_u0_b, d_st))
    for b in range(B):
        if b in bad:
            assert st_m[b] == capi.ERR_INVALID_TWIST and np.array_equal(ut_m[b], ut_bad[b]) and (u0_m[:, b] == -7.0).all()
        else:
            assert st_m[b] == 0 and np.array_equal(ut_m[b], ut_r[b]) and np.array_equal(u0_m[:, b], u0_r[:, b])
    # (b) non-finite poses
    p2 = poses.copy()
    p2[2, 0] = np.nan
    p2[A + 1, 1] = np.inf
    ut_b, u0_b, st_b = run(p2, ut0)
    for b in range(B):
        if b in (2, A + 1):
            assert np.isnan(u0_b[b, 0]) and np.isnan(ut_b[b, :, 0]).all()
        else:
            assert np.array_equal(ut_b[b], ut_ref[b]) and np.array_equal(u0_b[b], u0_ref[b])
    eng.close()


@pytest.mark.parametrize("L,steps", [(8, 20), (16, 50), (32, 50)])
def test_packed_consensus_ck_and_rollout(lanes, L, steps):
    """d_ck_shared (decentralised consensus: the shared c_k replaces the agent's own, README ref. [2]) against the oracle's
    switch, and eea_rollout_batch (optTraj, ergodic_control.hpp:313-317) on the packed kernel"""
    lanes(L)
    A = 64 // L
    B = A + 2
    eng, ors = make_pair("omni", 10, steps * 0.1, n_oracles=B)
    assert eng.agent_lanes(B) == L
    T, K2 = eng.T, eng.K2
    rng = np.random.default_rng(31 + L)
    poses = random_poses(rng, B)
    ut0 = rng.uniform(-0.5, 0.5, (B, T, 3))
    shared = rng.uniform(-0.05, 0.05, K2)
    shared[0] = 1.0
    d_ut, d_u0 = dev(ut0), torch.empty((B, 3), dtype=torch.float64, device="cuda")
    eng.control_batch(B, dev(poses), d_ut, d_u0, ck_shared=dev(shared))
    torch.cuda.synchronize()
    ut, u0 = d_ut.cpu().numpy(), d_u0.cpu().numpy()
    for b in range(B):
        ors[b].ut = ut0[b].T
        ors[b].set_shared_ck(shared)
        u, st = ors[b].control(MAP_BOUNDS, poses[b], None, stages=True)
        bar = TOL * max(1.0, np.abs(st["rhot"]).max())
        assert np.abs(ut[b].T - st["ut"]).max() <= bar and np.abs(u0[b] - u).max() <= bar
    # rollout of the updated controls == the oracle's optTraj
    d_traj = torch.empty((B, T, 3), dtype=torch.float64, device="cuda")
    eng.rollout_batch(B, dev(poses), d_ut, d_traj)
    torch.cuda.synchronize()
    traj = d_traj.cpu().numpy()
    for b in range(B):
        ref = ors[b].opt_traj()
        assert np.abs(traj[b].T[:2] - ref[:2]).max() <= TOL * 10
        d = traj[b].T[2] - ref[2]
        assert np.abs((d + np.pi) % (2 * np.pi) - np.pi).max() <= TOL * 10
    eng.close()


def test_engine_choice_of_lanes_per_agent():
    """automatic choice (EEA_OPT_AGENT_LANES = 0, cost model of csrc/control_pack_kernel.hip): packing only while the batch
    still fills the SIMDs; ineligible shapes (fp32, other K, long horizons) keep one wavefront per agent; the forced values
    fall back the same way"""
    capi.set_option(capi.OPT_AGENT_LANES, 0)
    # (round 6: refitted to the instances that run four wavefronts per SIMD -- the choices below are the measured winners of
    # profiles/r06_ablation.txt item 11; the argument is the batch of ONE call = one of two concurrent agent groups)
    eng, _ = make_pair("simple_cart", 10, 2.0, n_oracles=0)       # BASELINE configs[1]: T = 20
    assert eng.agent_lanes(64) == 64 and eng.agent_lanes(256) == 64
    assert eng.agent_lanes(2048) == 32 and eng.agent_lanes(4096) == 16
    assert eng.agent_lanes(6144) == 16 and eng.agent_lanes(12288) == 8 and eng.agent_lanes(32768) == 8
    eng.close()
    eng, _ = make_pair("omni", 10, 5.0, n_oracles=0)              # yaml as shipped: T = 50
    assert eng.agent_lanes(2048) == 64 and eng.agent_lanes(4096) == 16 and eng.agent_lanes(6144) == 16 and eng.agent_lanes(1 << 20) == 16
    eng.close()
    eng, _ = make_pair("omni", 5, 0.5, n_oracles=0)               # BASELINE configs[0]: T = 5
    assert eng.agent_lanes(2048) == 16 and eng.agent_lanes(4096) == 16 and eng.agent_lanes(16384) == 8 and eng.agent_lanes(1 << 20) == 8
    eng.close()
    eng, _ = make_pair("omni", 10, 20.0, n_oracles=0)             # T = 200: never
    assert eng.agent_lanes(1 << 20) == 64
    eng.close()
    eng, _ = make_pair("omni", 12, 2.0, n_oracles=0)              # K = 12: generic instance
    assert eng.agent_lanes(1 << 20) == 64
    eng.close()
    eng, _ = make_pair("omni", 10, 2.0, n_oracles=0, precision=capi.PREC_F32)
    assert eng.agent_lanes(1 << 20) == 64
    eng.close()
    try:
        capi.set_option(capi.OPT_AGENT_LANES, 8)
        eng, _ = make_pair("omni", 10, 5.0, n_oracles=0)          # T = 50 > 32: not in groups of 8
        assert eng.agent_lanes(4096) == 64
        eng.close()
        capi.set_option(capi.OPT_AGENT_LANES, 64)
        eng, _ = make_pair("omni", 10, 2.0, n_oracles=0)
        assert eng.agent_lanes(1 << 20) == 64
        eng.close()
    finally:
        capi.set_option(capi.OPT_AGENT_LANES, 0)


@pytest.mark.parametrize("model,K,horizon,B", [("simple_cart", 10, 2.0, 4096), ("omni", 5, 0.5, 8192), ("omni", 10, 5.0, 16384)])
def test_automatic_choice_full_batches_against_oracle(model, K, horizon, B):
    """the batches bench.py's other_configs leg runs, with the engine's own choice of lanes per agent, against the oracle on
    a sample of agents spread over the batch (first / last wavefront included)"""
    capi.set_option(capi.OPT_AGENT_LANES, 0)
    rng = np.random.default_rng(B + K)
    pick = sorted(set([0, 1, 7, 8, 63, 64, B // 2, B - 65, B - 2, B - 1] + list(rng.integers(0, B, 12))))
    eng, ors = make_pair(model, K, horizon, n_oracles=len(pick))
    assert eng.agent_lanes(B) in (8, 16, 32)
    T = eng.T
    poses = random_poses(rng, B)
    ut0 = rng.uniform(-0.5, 0.5, (B, T, 3))
    if model == "simple_cart":
        ut0[:, :, 1] = 0.0
    d_pose, d_ut = dev(poses), dev(ut0)
    d_u0 = torch.empty((B, 3), dtype=torch.float64, device="cuda")
    prev = ut0
    for call in range(2):
        eng.control_batch(B, d_pose, d_ut, d_u0)
        torch.cuda.synchronize()
        ut, u0 = d_ut.cpu().numpy(), d_u0.cpu().numpy()
        for o, b in zip(ors, pick):
            o.ut = prev[b].T
            u, st = o.control(MAP_BOUNDS, poses[b], None, stages=True)
            bar = TOL * max(1.0, np.abs(st["rhot"]).max())
            assert np.abs(ut[b].T - st["ut"]).max() <= bar and np.abs(u0[b] - u).max() <= bar, (b, call)
        prev = ut
    eng.close()


@pytest.mark.parametrize("L,steps,K", [(8, 20, 10), (16, 50, 10), (32, 50, 10), (8, 5, 5)])
def test_packed_result_does_not_depend_on_the_position_in_the_batch(lanes, L, steps, K):
    """The same agent (pose, warm start, replay memory) at different places of a batch -- another lane group of its
    wavefront (another block of the matrix instructions), another wavefront, the ragged last wavefront -- gives bitwise the
    same c_k, controls and u0: the blocks of v_mfma_f64_4x4x4_4b and the segments of the scans are independent, and
    nothing an agent computes depends on its neighbours (size-independent property, as test_full_size_batch_properties
    has it for the wavefront-per-agent kernel)."""
    lanes(L)
    A = 64 // L
    B = 5 * A + 3
    eng, _ = make_pair("omni", K, steps * 0.1, n_oracles=0)
    assert eng.agent_lanes(B) == L
    T, K2, n_mem = eng.T, eng.K2, 11
    rng = np.random.default_rng(99)
    poses = random_poses(rng, B)
    ut0 = rng.uniform(-0.5, 0.5, (B, T, 3))
    mem = random_poses(rng, B * n_mem).reshape(B, n_mem, 3)
    twins = [0, 1, A - 1, A, 2 * A + 1, 5 * A, B - 1]   # every one a copy of agent 0
    for t in twins[1:]:
        poses[t], ut0[t], mem[t] = poses[0], ut0[0], mem[0]
    d_ut, d_u0 = dev(ut0), torch.empty((B, 3), dtype=torch.float64, device="cuda")
    d_ck = torch.empty((B, K2), dtype=torch.float64, device="cuda")
    for _ in range(2):
        eng.control_batch(B, dev(poses), d_ut, d_u0, mem_cols=dev(mem), n_mem=torch.full((B,), n_mem, dtype=torch.int32, device="cuda"),
                          mem_stride=n_mem, ck=d_ck)
    torch.cuda.synchronize()
    ut, u0, ck = d_ut.cpu().numpy(), d_u0.cpu().numpy(), d_ck.cpu().numpy()
    for t in twins[1:]:
        assert np.array_equal(ut[t], ut[0]) and np.array_equal(u0[t], u0[0]) and np.array_equal(ck[t], ck[0]), t
    assert (ck[:, 0] == 1.0).all()   # mode (0,0) of c_k is the mean of ones, exactly
    eng.close()


@pytest.mark.parametrize("seed", range(24))
def test_packed_random_shapes(lanes, seed):
    """seeded random draws of (lanes per agent, horizon, model, K, batch size, replay memory, time step) inside the packed
    kernel's domain, stage outputs on and off, against the oracle"""
    rng = np.random.default_rng(1000 + seed)
    L = int(rng.choice([8, 16, 32]))
    lanes(L)
    A = 64 // L
    steps = int(rng.integers(2, 4 * L + 1))
    model = "omni" if rng.uniform() < 0.5 else "simple_cart"
    K = int(rng.choice([5, 10]))
    B = int(rng.integers(1, 4 * A + 2))
    n_mem = int(rng.choice([0, 0, 3, 17, 64]))
    dt = float(rng.choice([0.1, 0.125, 0.05]))
    _check_lanes(model, K, steps, dt, L, B) if dt == 0.125 else None
    run_batch_vs_oracle(model, K, steps * dt + 1e-9 if dt != 0.125 else steps * dt, dt, B=B, n_mem=n_mem, calls=2, seed=seed,
                        stages=bool(seed % 2))
