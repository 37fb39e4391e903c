"""eea_control_batch_steps (ABI 4): n consecutive receding-horizon optimisations per agent in ONE launch.

Contract (include/ergodic_amd.h): bitwise the same d_ut / d_u0 as n separate eea_control_batch calls fed with the same
pose rows -- the agent's wavefront reads its own stored controls back instead of the host launching again.  Checked
(a) bitwise against the separate calls, for both control kernels (the workgroup-per-agent kernel is issued as n launches
by the engine), fixed pose and a pose sequence, with replay memory, fp64 and fp32; (b) against the ORACLE: a logged pose
sequence replayed through independent oracle controllers, one control() per row (reference ergodic_control.hpp:224-311
called once per tick by exploration.hpp:232), u0 of every step and the final warm-start matrix <= 1e-9 per step of
accumulated feedback.
"""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from ergodic_exploration_amd import capi
from tests.gpu_util import MAP_BOUNDS, make_pair, random_poses
from tests.test_gpu_control_parity import dev

pytestmark = pytest.mark.gpu


def _inputs(model, eng, B, n_steps, n_mem, seed, tdt=torch.float64):
    rng = np.random.default_rng(seed)
    T = eng.T
    pose0 = random_poses(rng, B)
    # a pose sequence: small random walk per step (what a tf lookup would deliver tick by tick)
    seq = pose0[None] + np.cumsum(rng.normal(scale=0.02, size=(n_steps, B, 3)), axis=0)
    ut0 = rng.uniform(-0.4, 0.4, (B, T, 3))
    if model == "simple_cart":
        ut0[:, :, 1] = 0.0
    mem = random_poses(rng, B * n_mem).reshape(B, n_mem, 3) if n_mem else None
    return seq, ut0, mem


@pytest.mark.parametrize("model,K,horizon,n_mem,precision", [
    ("simple_cart", 10, 20.0, 0, capi.PREC_F64),    # the metric point
    ("omni", 10, 19.5, 40, capi.PREC_F64),          # cooperative last slot + replay memory
    ("omni", 5, 0.5, 0, capi.PREC_F64),             # config 1
    ("omni", 20, 5.0, 0, capi.PREC_F32),            # config 3 shape, fp32
    ("omni", 12, 3.0, 3, capi.PREC_F64),            # generic K <= 16 instance
    ("omni", 30, 6.0, 0, capi.PREC_F64),            # workgroup-per-agent kernel: n launches inside the call
    ("omni", 7, 30.0, 0, capi.PREC_F64),            # T = 300 > 256: workgroup kernel
])
def test_steps_in_one_launch_equal_separate_calls(model, K, horizon, n_mem, precision):
    B, n_steps = 37, 5
    tdt = torch.float64 if precision == capi.PREC_F64 else torch.float32
    eng, _ = make_pair(model, K, horizon, n_oracles=0, precision=precision)
    seq, ut0, mem = _inputs(model, eng, B, n_steps, n_mem, seed=7)
    d_seq = dev(seq, tdt)
    d_mem = dev(mem, tdt) if n_mem else None
    d_nmem = torch.full((B,), n_mem, dtype=torch.int32, device="cuda") if n_mem else None
    kw = dict(mem_cols=d_mem, n_mem=d_nmem, mem_stride=n_mem)
    # (1) separate calls, one per pose row
    ut_a = dev(ut0, tdt)
    u0_a = torch.empty((n_steps, B, 3), dtype=tdt, device="cuda")
    for n in range(n_steps):
        eng.control_batch(B, d_seq[n], ut_a, u0_a[n], **kw)
    # (2) one launch, pose sequence and per-step u0 rows
    ut_b = dev(ut0, tdt)
    u0_b = torch.full((n_steps, B, 3), float("nan"), dtype=tdt, device="cuda")
    d_ck = torch.empty((B, eng.K2), dtype=tdt, device="cuda")
    eng.control_batch(B, d_seq, ut_b, u0_b, n_steps=n_steps, pose_step_stride=B, u0_step_stride=B, ck=d_ck, **kw)
    torch.cuda.synchronize()
    assert torch.equal(ut_a, ut_b) and torch.equal(u0_a, u0_b)
    # (3) fixed pose (stride 0): every step from row 0, only the last u0 kept
    ut_c, ut_d = dev(ut0, tdt), dev(ut0, tdt)
    u0_c = torch.empty((B, 3), dtype=tdt, device="cuda")
    u0_d = torch.empty((B, 3), dtype=tdt, device="cuda")
    for n in range(n_steps):
        eng.control_batch(B, d_seq[0], ut_c, u0_c, **kw)
    eng.control_batch(B, d_seq[0], ut_d, u0_d, n_steps=n_steps, **kw)
    torch.cuda.synchronize()
    assert torch.equal(ut_c, ut_d) and torch.equal(u0_c, u0_d)
    eng.close()


@pytest.mark.parametrize("L,model,horizon", [(8, "simple_cart", 2.0), (16, "omni", 5.0), (32, "omni", 5.0)])
def test_packed_steps_in_one_launch_leave_the_last_steps_records(L, model, horizon):
    """several agents per wavefront, n_steps > 1: c_k and the sum records (one per agent, one per wavefront) of the launch are those
    of its LAST step -- bitwise what the last of the separate calls leaves"""
    capi.set_option(capi.OPT_AGENT_LANES, L)
    try:
        A = 64 // L
        B, n_steps = 9 * A + 1, 4
        eng, _ = make_pair(model, 10, horizon, n_oracles=0)
        assert eng.agent_lanes(B) == L
        seq, ut0, _ = _inputs(model, eng, B, n_steps, 0, seed=11)
        d_seq = dev(seq)
        RL, K2, W = eng.ck_record_len, eng.K2, eng.record_count(B)
        out = {}
        for form in ("calls", "launch"):
            ut, u0 = dev(ut0), torch.empty((n_steps, B, 3), dtype=torch.float64, device="cuda")
            ck = torch.zeros((B, K2), dtype=torch.float64, device="cuda")
            arec = torch.zeros((B, RL), dtype=torch.float64, device="cuda")
            wrec = torch.zeros((W, RL), dtype=torch.float64, device="cuda")
            for wave, rec in ((False, arec), (True, wrec)):
                utw = ut.clone()
                if form == "calls":
                    for n in range(n_steps):
                        eng.control_batch(B, d_seq[n], utw, u0[n], ck=ck, ck_rec=rec, rec_per_wavefront=wave)
                else:
                    eng.control_batch(B, d_seq, utw, u0, n_steps=n_steps, pose_step_stride=B, u0_step_stride=B, ck=ck, ck_rec=rec,
                                      rec_per_wavefront=wave)
            torch.cuda.synchronize()
            out[form] = (utw, u0, ck, arec, wrec)
        for a, b in zip(out["calls"], out["launch"]):
            assert torch.equal(a, b)
        assert float(out["launch"][4][:, K2].sum()) == B and float(out["launch"][3][:, K2].sum()) == B
        eng.close()
    finally:
        capi.set_option(capi.OPT_AGENT_LANES, 0)


def test_steps_argument_errors():
    eng, _ = make_pair("omni", 10, 2.0, n_oracles=0)
    B, T = 4, eng.T
    d_pose = dev(random_poses(np.random.default_rng(0), B))
    d_ut = torch.zeros((B, T, 3), dtype=torch.float64, device="cuda")
    d_u0 = torch.empty((B, 3), dtype=torch.float64, device="cuda")
    for bad in (dict(n_steps=0), dict(n_steps=2, pose_step_stride=B - 1), dict(n_steps=2, u0_step_stride=1)):
        with pytest.raises(capi.EngineError) as ei:
            eng.control_batch(B, d_pose, d_ut, d_u0, **bad)
        assert ei.value.status == capi.ERR_INVALID_ARGUMENT
    eng.close()


@pytest.mark.parametrize("model,K,horizon,n_mem", [("simple_cart", 10, 20.0, 0), ("omni", 10, 5.0, 7), ("omni", 5, 19.8, 0)])
def test_pose_sequence_replay_against_oracle(model, K, horizon, n_mem):
    """a logged pose sequence through the controller, one launch for all its ticks, against independent oracle
    controllers called once per tick.  The oracle is re-seeded with the kernel's controls after every tick would need a
    read-back per step -- instead the whole closed recursion is compared: differences feed back through the warm start,
    so the bar grows with the tick count (1e-9 x 10 per tick, as the dependent-call tests of round 3)."""
    B, n_steps = 4, 4
    eng, ors = make_pair(model, K, horizon, n_oracles=B)
    seq, ut0, mem = _inputs(model, eng, B, n_steps, n_mem, seed=19)
    d_mem = dev(mem) if n_mem else None
    d_nmem = torch.full((B,), n_mem, dtype=torch.int32, device="cuda") if n_mem else None
    d_ut = dev(ut0)
    d_u0 = torch.empty((n_steps, B, 3), dtype=torch.float64, device="cuda")
    eng.control_batch(B, dev(seq), d_ut, d_u0, n_steps=n_steps, pose_step_stride=B, u0_step_stride=B,
                      mem_cols=d_mem, n_mem=d_nmem, mem_stride=n_mem)
    torch.cuda.synchronize()
    u0, ut = d_u0.cpu().numpy(), d_ut.cpu().numpy()
    for b in range(B):
        ors[b].ut = ut0[b].T
        for n in range(n_steps):
            u = ors[b].control(MAP_BOUNDS, seq[n, b], mem[b].T if n_mem else None)
            assert np.abs(u0[n, b] - u).max() <= 1e-9 * 10 ** n, (b, n, u0[n, b], u)
        assert np.abs(ut[b].T - ors[b].ut).max() <= 1e-9 * 10 ** n_steps
    eng.close()
