"""GPU parity of the fused control kernel (through the C ABI) against the CPU oracle.

Tolerances: c_k <= 1e-11 abs; trajectory, co-state, gradients and
controls <= 1e-9 (the kernel re-associates the RK4 sums into scans and evaluates the
separable basis by recurrence); headings compared modulo 2 pi.  fp32: <= 1e-4 on u.
Every bar is |delta| <= tol * max(1, max |oracle stage|): absolute while a stage is of unit size,
relative to the stage's own magnitude where it is not (a robot that leaves the map drives the barrier
gradient and the co-state to 1e3-1e4); a stage of unit size never gets a looser bar.
"""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from oracle import pyoracle as po
from ergodic_exploration_amd import capi
from tests.gpu_util import MAP_BOUNDS, MEANS, SIGMAS, angle_diff, make_pair, random_poses

pytestmark = pytest.mark.gpu
MODELS_PO = {"omni": po.MODEL_OMNI, "simple_cart": po.MODEL_SIMPLE_CART}

TOL_CK = 1e-11
TOL = 1e-9


def dev(a, dtype=torch.float64):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).cuda()


def run_batch_vs_oracle(model, K, horizon, dt, B, n_mem, calls, seed, precision=capi.PREC_F64,
                        tol=TOL, tol_ck=TOL_CK, bounds=MAP_BOUNDS, means=MEANS, sigmas=SIGMAS, tol_u_rho=0.0,
                        stages=True):
    """stages=False: no per-stage output pointers are passed, which selects the kernel instances compiled WITHOUT the
    stage outputs -- the ones bench.py times (STAGES = false, control_wave_kernel.hip); c_k,
    the warm-start matrix ut (every step's control: the view of the co-state that is left) and u0 are compared, on the
    same bars as with stages.
    tol_u_rho (fp32 runs only): u = clamp(-Rinv B^T rho) inherits the co-state's ABSOLUTE rounding error wherever
    it is not clamped, so when the co-state is far from unit size (robot outside the map: |rho| ~ 1e3) the bar on the
    controls is max(tol max(1, |u|), tol_u_rho |rho|max); fp64 runs keep the pure per-stage bar (tol_u_rho = 0)."""
    rng = np.random.default_rng(seed)
    eng, ors = make_pair(model, K, horizon, dt=dt, precision=precision, n_oracles=B, bounds=bounds,
                         means=means, sigmas=sigmas)
    T, K2 = eng.T, eng.K2
    tdt = torch.float64 if precision == capi.PREC_F64 else torch.float32
    poses = random_poses(rng, B, bounds)
    # warm-start controls: random but inside the limits (SimpleCart: vy = 0)
    ut0 = rng.uniform(-0.5, 0.5, (B, T, 3))
    if model == "simple_cart":
        ut0[:, :, 1] = 0.0
    mem = None
    if n_mem:
        mem = random_poses(rng, B * n_mem, bounds).reshape(B, n_mem, 3)
    d_pose, d_ut = dev(poses, tdt), dev(ut0, tdt)
    d_u0 = torch.empty((B, 3), dtype=tdt, device="cuda")
    outs = {k: torch.empty((B, T, 3), dtype=tdt, device="cuda") for k in ("traj", "edx", "bdx", "rhot")} if stages else {}
    d_ck = torch.empty((B, K2), dtype=tdt, device="cuda")
    d_status = torch.full((B,), -1, dtype=torch.int32, device="cuda")
    d_mem = dev(mem, tdt) if n_mem else None
    d_nmem = torch.full((B,), n_mem, dtype=torch.int32, device="cuda") if n_mem else None
    for b in range(B):
        ors[b].ut = ut0[b].T
    worst, scaled, magn = {}, {}, {}   # absolute differences, differences / max(1, |stage|), stage magnitudes
    for call in range(calls):
        eng.control_batch(B, d_pose, d_ut, d_u0, mem_cols=d_mem, n_mem=d_nmem, mem_stride=n_mem,
                          ck=d_ck, status=d_status, **outs)
        torch.cuda.synchronize()
        assert (d_status.cpu().numpy() == 0).all()
        got = {k: v.cpu().numpy().astype(np.float64) for k, v in outs.items()}
        got["ck"] = d_ck.cpu().numpy().astype(np.float64)
        got["ut"] = d_ut.cpu().numpy().astype(np.float64)
        got["u0"] = d_u0.cpu().numpy().astype(np.float64)
        for b in range(B):
            u, st = ors[b].control(bounds, poses[b], mem[b].T if n_mem else None, stages=True)
            errs = {
                "ck": (np.abs(got["ck"][b] - st["ck"]).max(), np.abs(st["ck"]).max()),
                "ut": (np.abs(got["ut"][b].T - st["ut"]).max(), np.abs(st["ut"]).max()),
                "u0": (np.abs(got["u0"][b] - u).max(), np.abs(u).max()),
            }
            if stages:
                errs.update({
                    "traj_xy": (np.abs(got["traj"][b].T[:2] - st["traj"][:2]).max(), np.abs(st["traj"][:2]).max()),
                    "traj_th": (np.abs(angle_diff(got["traj"][b].T[2], st["traj"][2])).max(), 1.0),
                    "edx": (np.abs(got["edx"][b].T - st["edx"]).max(), np.abs(st["edx"]).max()),
                    "bdx": (np.abs(got["bdx"][b].T - st["bdx"]).max(), np.abs(st["bdx"]).max()),
                    "rhot": (np.abs(got["rhot"][b].T - st["rhot"]).max(), np.abs(st["rhot"]).max()),
                })
            else:   # the co-state's magnitude still scales the bar of the controls it drives (from the oracle's stage)
                errs["rhot"] = (0.0, np.abs(st["rhot"]).max())
            for k, (v, mag) in errs.items():
                worst[k] = max(worst.get(k, 0.0), float(v))
                scaled[k] = max(scaled.get(k, 0.0), float(v) / max(1.0, float(mag)))
                magn[k] = max(magn.get(k, 0.0), float(mag))
            # feed the oracle's controls forward from the kernel's so both start each call
            # from identical state (SURVEY.md section 7 "hard parts": never compare long closed loops)
            ors[b].ut = got["ut"][b].T
    eng.close()
    if os.environ.get("EEA_PRINT_WORST"):
        print("worst", model, K, T, n_mem, "dt=%g" % dt, "f32" if precision == capi.PREC_F32 else "f64",
              "stages" if stages else "NO-STAGES(timed instance)",
              {k: "%.2e (|stage| %.1e)" % (worst[k], magn[k]) for k in worst})
    assert scaled["ck"] <= tol_ck, (worst, magn)
    for k in ("traj_xy", "traj_th", "edx", "bdx", "rhot") if stages else ():
        assert scaled[k] <= tol, (k, worst, magn)
    for k in ("ut", "u0"):
        assert scaled[k] <= tol or worst[k] <= tol_u_rho * magn["rhot"], (k, worst, magn)
    return worst


@pytest.mark.parametrize("model,K,horizon,dt,n_mem", [
    ("omni", 5, 0.5, 0.1, 0),            # BASELINE config 1 (synthetic small)
    ("simple_cart", 10, 2.0, 0.1, 0),    # BASELINE config 2
    ("omni", 10, 5.0, 0.1, 7),           # yaml as shipped, memory <= batch
    ("simple_cart", 10, 5.0, 0.1, 100),  # yaml as shipped, full memory batch
    ("omni", 10, 20.0, 0.1, 0),          # metric point K=10 T=200
    ("simple_cart", 10, 20.0, 0.1, 100), # metric point with memory
])
def test_stagewise_parity_f64(model, K, horizon, dt, n_mem):
    run_batch_vs_oracle(model, K, horizon, dt, B=6, n_mem=n_mem, calls=3, seed=11)


def test_generic_basis_count_and_long_horizon():
    # K = 7 takes the runtime-K kernel; T = 300 > 256 exercises the chunked scans
    run_batch_vs_oracle("omni", 7, 30.0, 0.1, B=3, n_mem=5, calls=2, seed=5)
    run_batch_vs_oracle("simple_cart", 12, 3.0, 0.1, B=3, n_mem=0, calls=2, seed=6)


@pytest.mark.parametrize("steps", [63, 64, 65, 100, 128, 129])
def test_wavefronts_per_agent_boundaries(steps):
    """Horizons of <= 64 / <= 128 steps run on one / two wavefronts per agent (control_kernel.hip
    control_threads); dt = 0.125 keeps horizon / dt exact around the boundaries."""
    run_batch_vs_oracle("omni", 10, steps * 0.125, 0.125, B=3, n_mem=9, calls=2, seed=21)
    run_batch_vs_oracle("simple_cart", 6, steps * 0.125, 0.125, B=2, n_mem=0, calls=2, seed=22)


@pytest.mark.parametrize("steps", [191, 192, 193, 255, 256, 257])
def test_steps_per_lane_boundaries(steps):
    """The wavefront-per-agent kernel gives a lane ceil(T / 64) <= 4 consecutive steps: horizons around the
    3 -> 4 steps-per-lane boundary, the last eligible horizon (256) and the first one that falls back to the
    workgroup kernel (257); both models, with replay memory longer than one 64-point pass; K = 16 / 17 straddle
    the one- / two-tile contraction and the eligibility rule (K <= 16 or K = 20)."""
    run_batch_vs_oracle("omni", 10, steps * 0.125, 0.125, B=5, n_mem=70, calls=2, seed=41)
    run_batch_vs_oracle("simple_cart", 16, steps * 0.125, 0.125, B=2, n_mem=0, calls=2, seed=42)
    run_batch_vs_oracle("omni", 17, steps * 0.125, 0.125, B=2, n_mem=3, calls=1, seed=43)
    run_batch_vs_oracle("omni", 20, steps * 0.125, 0.125, B=2, n_mem=65, calls=2, seed=44)


@pytest.mark.parametrize("steps", [2, 3, 4, 29, 32, 33, 36, 37, 66, 72, 73, 125, 129, 140, 141, 217])
def test_contraction_row_group_boundaries(steps):
    """The contraction of the wavefront kernel stages a pass of 32 points with ALL 64 lanes (one axis per lane, the
    partner point's cosine over v_permlane32_swap) and skips row groups of 4 points past the horizon: horizons whose
    number of valid lanes ceil((T - j) / S) sits at / next to 32 and to multiples of 4 (one, two and four steps per
    lane; the two-step minimum of the reference; the replay-memory pass with fewer than 32 / more than 32 columns), K = 10 and 5
    (software-pipelined pass), K = 12 (generic instance), fp64; K = 10 again in fp32."""
    run_batch_vs_oracle("simple_cart", 10, steps * 0.125, 0.125, B=3, n_mem=0, calls=2, seed=61)
    run_batch_vs_oracle("omni", 5, steps * 0.125, 0.125, B=2, n_mem=30, calls=2, seed=62)
    run_batch_vs_oracle("omni", 12, steps * 0.125, 0.125, B=2, n_mem=35, calls=1, seed=63)
    # fp32 against the fp64 oracle (<= 5e-4 on the co-state; every bar relative to max(1, |stage|): with
    # dt = 0.125 the longer horizons leave the 12 m map and the barrier gradient / co-state reach 1e2-1e3)
    run_batch_vs_oracle("omni", 10, steps * 0.125, 0.125, B=2, n_mem=33, calls=2, seed=64, precision=capi.PREC_F32,
                        tol=5e-4, tol_ck=1e-5, tol_u_rho=2e-6)


@pytest.mark.parametrize("steps", [66, 72, 129, 130, 136, 193, 194, 197, 199, 200])
def test_top_heavy_horizons_and_cooperative_last_slot(steps):
    """T = 64 (S - 1) + r with r <= 8: the first r lanes of the agent's wavefront own S steps, all others S - 1 (full
    slots + one nearly empty last slot).  With four slots (T = 193 .. 200, the metric's T = 200 among them) the fp64
    K = 5 / 10 instances take the gradient of the last slot's r steps with all 64 lanes, 8 per step (other summation
    order, rows' cos / sin by binary powering).  Both models, with and without replay memory, two calls (the second on
    the first one's controls); r = 1, 2, 5, 7, 8 and the shapes with two / three slots that keep the regular pass."""
    run_batch_vs_oracle("simple_cart", 10, steps * 0.1, 0.1, B=4, n_mem=0, calls=2, seed=71)
    run_batch_vs_oracle("omni", 10, steps * 0.1, 0.1, B=3, n_mem=40, calls=2, seed=72)
    run_batch_vs_oracle("omni", 5, steps * 0.1, 0.1, B=3, n_mem=0, calls=2, seed=73)
    run_batch_vs_oracle("simple_cart", 12, steps * 0.1, 0.1, B=2, n_mem=0, calls=1, seed=74)   # generic instance: regular pass


@pytest.mark.parametrize("dt", [0.1, 1.0, 2.0])
def test_small_and_large_step_increments(dt):
    """The wavefront kernel takes the sin/cos of a step's mid-stage / post-step heading and of its later basis
    angles by rotating the previous values with a short series of the INCREMENT when every increment of the agent
    is small (|dt w / 2| <= pi/16, |dx| <= lx/16), and evaluates them in full otherwise: dt = 0.1 stays on the
    rotation path, dt = 1 leaves it for the headings (warm-start yaw rates up to 0.5), dt = 2 also for the basis
    angles (steps of up to 1 m on the 12 m map).  Same bars on both paths."""
    # at dt >= 1 the robot leaves the 12 m map within a few steps: the barrier gradient 2 x 25 x distance and with it
    # the co-state reach ~1e3-1e4; the bar stays 1e-9 x max(1, |stage|) (run_batch_vs_oracle), no looser constant
    run_batch_vs_oracle("omni", 10, 40 * dt, dt, B=4, n_mem=5, calls=2, seed=51)
    run_batch_vs_oracle("simple_cart", 10, 70 * dt, dt, B=3, n_mem=0, calls=2, seed=52)
    if dt == 0.1:
        run_batch_vs_oracle("simple_cart", 10, 200 * dt, dt, B=3, n_mem=0, calls=2, seed=53)


@pytest.mark.parametrize("block", [64, 128])
def test_forced_threads_per_agent_long_horizon(block):
    """EEA_OPT_WORKGROUP_THREADS forces fewer threads per agent than horizon steps on the workgroup-per-agent
    kernel (EEA_OPT_CONTROL_KERNEL = 1): several steps per lane through the chunk loop."""
    capi.set_option(capi.OPT_CONTROL_KERNEL, 1)
    capi.set_option(capi.OPT_WORKGROUP_THREADS, block)
    try:
        assert capi.get_option(capi.OPT_WORKGROUP_THREADS) == block
        run_batch_vs_oracle('simple_cart', 10, 20.0, 0.1, B=3, n_mem=40, calls=2, seed=31)
        run_batch_vs_oracle('omni', 7, 30.0, 0.1, B=2, n_mem=0, calls=2, seed=32)
    finally:
        capi.set_option(capi.OPT_CONTROL_KERNEL, 0)
        capi.set_option(capi.OPT_WORKGROUP_THREADS, 0)


def test_config3_shape_f64_and_f32():
    # BASELINE config 3: Omni, K = 20, dt 0.02, horizon 5 (T = 250), 256 x 256 target grid
    bounds = (0.0, 25.5, 0.0, 25.5)
    means, sigmas = [[6.0, 6.0], [19.0, 12.0]], [[3.0, 3.0], [3.0, 3.0]]
    run_batch_vs_oracle("omni", 20, 5.0, 0.02, B=2, n_mem=0, calls=2, seed=3, bounds=bounds,
                        means=means, sigmas=sigmas)
    # fp32 engine against the fp64 oracle: <= 1e-4 on controls (asserted on its own below), 5e-4 on the co-state
    worst = run_batch_vs_oracle("omni", 20, 5.0, 0.02, B=2, n_mem=0, calls=2, seed=3, bounds=bounds,
                                means=means, sigmas=sigmas, precision=capi.PREC_F32, tol=5e-4, tol_ck=1e-5)
    assert worst["u0"] <= 1e-4 and worst["ut"] <= 1e-4, worst


def test_k30_t500_shape():
    # BASELINE config 5 control shape (K = 30, T = 500) on the Gaussian target
    run_batch_vs_oracle("omni", 30, 50.0, 0.1, B=1, n_mem=0, calls=1, seed=8)


@pytest.mark.parametrize("steps,n_mem", [(2, 0), (7, 3), (64, 0), (70, 61), (129, 100), (300, 37)])
def test_k30_cooperative_staging(steps, n_mem):
    """K = 30 stages the contraction tiles with all 64 lanes (point x axis x group of 8 modes, cos / sin of 8, 16, 24 times
    the angle by doubling): horizons below / at / above one pass of 8 points and one wavefront, one and two chunks,
    replay memory before the rollout points (ragged last passes), both models, warm-started second call."""
    run_batch_vs_oracle("omni", 30, steps * 0.125, 0.125, B=3, n_mem=n_mem, calls=2, seed=71)
    run_batch_vs_oracle("simple_cart", 30, steps * 0.125, 0.125, B=2, n_mem=n_mem, calls=1, seed=72)


def test_survey_anchors_through_c_abi(anchors):
    """End-to-end control() outputs of the reference's own sources through
    the single-agent host entry point eea_control, closed loop."""
    c = anchors["closed_loop_common"]
    for key in ("omni_K10_T50", "simple_cart_K10_T20"):
        a = anchors[key]
        cm = {"omni": capi.MODEL_OMNI, "simple_cart": capi.MODEL_SIMPLE_CART}[a["model"]]
        om = {"omni": po.MODEL_OMNI, "simple_cart": po.MODEL_SIMPLE_CART}[a["model"]]
        lim = np.array(a["limits"])
        eng = capi.Engine(capi.make_config(cm, c["dt"], a["horizon"], c["target_resolution"],
                                           c["expl_weight"], a["num_basis"], np.diag(a["Rinv_diag"]),
                                           -lim, lim))
        eng.set_target_gaussians(c["means"], c["sigmas"])
        x = np.array(c["x0"])
        for i, exp in enumerate(a["u"]):
            u = eng.control(c["map_bounds"], x)
            assert np.abs(u - np.array(exp)).max() < 1e-9 * (10 ** i), (key, i, u, exp)
            st, x = po.rk4_step_fwd(om, c["dt"], x, np.array(exp))  # drive with the reference's u
            assert st == po.OK
        eng.close()

    a = anchors["memory_omni_K5"]
    lim = np.array(a["limits"])
    eng = capi.Engine(capi.make_config(capi.MODEL_OMNI, a["dt"], a["horizon"], a["target_resolution"],
                                       1.0, a["num_basis"], np.diag(a["Rinv_diag"]), -lim, lim))
    eng.set_target_gaussians(a["means"], a["sigmas"])
    u = eng.control(a["map_bounds"], a["x"], np.array(a["memory"]).T)
    assert np.abs(u - np.array(a["u"])).max() < 1e-9
    eng.close()


def test_single_agent_state_and_opt_traj():
    """eea_control keeps ut_ in the engine (warm start), eea_opt_traj == optTraj()."""
    eng, (orc,) = make_pair("omni", 10, 5.0)
    x = np.array([1.0, 1.0, 0.3])
    for _ in range(3):
        u = eng.control(MAP_BOUNDS, x)
        uo = orc.control(MAP_BOUNDS, x)
        assert np.abs(u - uo).max() < 1e-8
        assert np.abs(eng.get_ut() - orc.ut).max() < 1e-8
        tr, tro = eng.opt_traj(), orc.opt_traj()
        assert np.abs(tr[:2] - tro[:2]).max() < 1e-8
        assert np.abs(angle_diff(tr[2], tro[2])).max() < 1e-8
        orc.ut = eng.get_ut()
        st, x = po.rk4_step_fwd(po.MODEL_OMNI, 0.1, x, u)
    eng.close()


def test_simple_cart_invalid_twist_status():
    """SimpleCart::operator() throws on |u1| >= 1e-12 (cart.hpp:167-170): per-agent status,
    the offending agent's buffers untouched, the others unaffected."""
    eng, _ = make_pair("simple_cart", 10, 2.0, n_oracles=0)
    B, T = 3, eng.T
    rng = np.random.default_rng(0)
    ut = rng.uniform(-0.3, 0.3, (B, T, 3))
    ut[:, :, 1] = 0.0
    ut[1, 5, 1] = 1e-3
    d_ut = dev(ut)
    d_pose = dev(random_poses(rng, B))
    d_u0 = torch.zeros((B, 3), dtype=torch.float64, device="cuda")
    d_status = torch.full((B,), -1, dtype=torch.int32, device="cuda")
    eng.control_batch(B, d_pose, d_ut, d_u0, status=d_status)
    torch.cuda.synchronize()
    assert d_status.cpu().tolist() == [0, capi.ERR_INVALID_TWIST, 0]
    assert np.array_equal(d_ut.cpu().numpy()[1], ut[1])
    eng.set_ut(ut[1].T)
    with pytest.raises(capi.EngineError) as ei:
        eng.control(MAP_BOUNDS, [1.0, 1.0, 0.0])
    assert ei.value.status == capi.ERR_INVALID_TWIST
    eng.close()


def test_constructor_errors():
    with pytest.raises(capi.EngineError) as ei:  # horizon == dt -> steps == 1 (ergodic_control.hpp:212-216)
        capi.Engine(capi.make_config(capi.MODEL_OMNI, 0.1, 0.1, 0.1, 1.0, 5, np.eye(3), [-1] * 3, [1] * 3))
    assert ei.value.status == capi.ERR_INVALID_ARGUMENT
    with pytest.raises(capi.EngineError):        # Cart / Mecanum are not usable with ErgodicControl
        capi.Engine(capi.make_config(2, 0.1, 1.0, 0.1, 1.0, 5, np.eye(3), [-1] * 3, [1] * 3))
    eng = capi.Engine(capi.make_config(capi.MODEL_OMNI, 0.1, 1.0, 0.1, 1.0, 5, np.eye(3), [-1] * 3, [1] * 3))
    with pytest.raises(capi.EngineError) as ei:  # control before any target
        eng.control(MAP_BOUNDS, [0, 0, 0])
    assert ei.value.status == capi.ERR_NO_TARGET
    eng.close()


def test_batch_without_target_is_an_error_not_nan():
    """Documented difference from the reference: ErgodicControl with a default Target (setTarget skipped) divides 0 / 0 in
    Target::fill (target.cpp:87) and control() returns NaN (ergodic_control.hpp:411-413, SURVEY hazard 12); the engine
    returns EEA_ERR_NO_TARGET from every control entry instead and touches nothing.  The NaN itself is still reachable the
    reference's way: an explicit EMPTY Gaussian list is a target whose grid is 0 / 0."""
    eng = capi.Engine(capi.make_config(capi.MODEL_OMNI, 0.1, 1.0, 0.1, 1.0, 5, np.eye(3), [-1] * 3, [1] * 3))
    B, T = 3, eng.T
    d_pose = dev(random_poses(np.random.default_rng(1), B))
    d_ut = torch.full((B, T, 3), 0.25, dtype=torch.float64, device="cuda")
    d_u0 = torch.full((B, 3), -7.0, dtype=torch.float64, device="cuda")
    for call in (lambda: eng.control_batch(B, d_pose, d_ut, d_u0),
                 lambda: eng.control_batch(B, d_pose, d_ut, d_u0, n_steps=2),
                 lambda: eng.control(MAP_BOUNDS, [0.0, 0.0, 0.0])):
        with pytest.raises(capi.EngineError) as ei:
            call()
        assert ei.value.status == capi.ERR_NO_TARGET
    torch.cuda.synchronize()
    assert (d_ut.cpu().numpy() == 0.25).all() and (d_u0.cpu().numpy() == -7.0).all()
    # the reference's NaN, by the reference's route: a target without Gaussians
    eng.set_target_gaussians(np.zeros((0, 2)), np.zeros((0, 2)))
    eng.config_domain(MAP_BOUNDS)
    eng.control_batch(B, d_pose, d_ut, d_u0)
    torch.cuda.synchronize()
    assert np.isnan(d_u0.cpu().numpy()).all()
    eng.close()


def _wrapped(a):
    return np.array([po.normalize_angle_PI(v) for v in np.ravel(a)]).reshape(np.shape(a))


@pytest.mark.parametrize("model,K,steps,lanes", [("omni", 10, 200, 0), ("simple_cart", 10, 20, 0), ("omni", 10, 20, 8),
                                                 ("simple_cart", 5, 50, 16), ("omni", 30, 70, 0)])
def test_headings_of_the_hip_path_lie_in_minus_pi_pi(model, K, steps, lanes):
    """normalize_angle_PI (numerics.hpp:77-89) wraps to [-pi, pi): pi -> -pi.  The stage-wise tests compare headings modulo
    2 pi only; here the CONVENTION is pinned on the HIP path's own outputs (d_traj of a control call, eea_rollout_batch,
    eea_opt_traj), on every kernel (wavefront per agent, several agents per wavefront, workgroup per agent):
    (a) poses whose heading is exactly pi, -pi, 3 pi, 7, -7 with zero controls: every reported heading is bitwise
        -pi / the oracle's normalize_angle_PI value to 4e-16 |theta_0|;
    (b) fast spins (|w| up to the limit 2 rad/s over the horizon: several turns): -pi <= theta < pi everywhere, and equal to
        the oracle's rollout WITHOUT a modulo wherever the oracle is more than 1e-6 away from the cut."""
    capi.set_option(capi.OPT_AGENT_LANES, lanes)
    try:
        dt = 0.1
        eng, ors = make_pair(model, K, steps * dt, dt=dt, n_oracles=1)
        T = eng.T
        th0 = np.array([np.pi, -np.pi, 3 * np.pi, 7.0, -7.0, 0.0, np.nextafter(np.pi, 0.0), -3 * np.pi])
        B = th0.size
        poses = np.stack([np.full(B, 3.0), np.full(B, 2.0), th0], 1)
        d_ut = torch.zeros((B, T, 3), dtype=torch.float64, device="cuda")
        d_traj = torch.empty((B, T, 3), dtype=torch.float64, device="cuda")
        want = _wrapped(th0)
        assert want[0] == -np.pi and want[1] == -np.pi and want[2] == -np.pi   # (the oracle's own anchors)
        eng.rollout_batch(B, dev(poses), d_ut, d_traj)
        torch.cuda.synchronize()
        th = d_traj.cpu().numpy()[:, :, 2]
        for b in range(B):
            if want[b] == -np.pi:
                assert (th[b] == -np.pi).all(), (b, th[b][:3])
            else:
                assert np.abs(th[b] - want[b]).max() <= 4e-16 * max(1.0, abs(th0[b])), (b, th[b][:3], want[b])
        # ... and the trajectory a control call reports (warm start zero: the shifted controls are zero as well)
        d_u0 = torch.empty((B, 3), dtype=torch.float64, device="cuda")
        d_traj.fill_(9.0)
        eng.control_batch(B, dev(poses), d_ut, d_u0, traj=d_traj)
        torch.cuda.synchronize()
        th = d_traj.cpu().numpy()[:, :, 2]
        assert (th[want == -np.pi] == -np.pi).all()
        # single-agent entry (workgroup kernel): eea_opt_traj after a control() from heading pi
        eng.set_ut(np.zeros((3, T)))
        eng.control(MAP_BOUNDS, [3.0, 2.0, np.pi])
        eng.set_ut(np.zeros((3, T)))
        assert (eng.opt_traj()[2] == -np.pi).all()
        # (b) fast spins
        rng = np.random.default_rng(17)
        B = 6
        poses = random_poses(rng, B)
        ut = np.zeros((B, T, 3))
        ut[:, :, 0] = rng.uniform(-0.3, 0.3, (B, T))
        ut[:, :, 2] = rng.uniform(1.0, 2.0, (B, 1)) * np.where(rng.uniform(size=(B, 1)) < 0.5, -1.0, 1.0)
        d_traj = torch.empty((B, T, 3), dtype=torch.float64, device="cuda")
        eng.rollout_batch(B, dev(poses), dev(ut), d_traj)
        torch.cuda.synchronize()
        th = d_traj.cpu().numpy()[:, :, 2]
        assert (th >= -np.pi).all() and (th < np.pi).all()
        for b in range(B):
            st, xt = po.rk4_solve_fwd(MODELS_PO[model], dt, steps * dt, poses[b], ut[b].T)
            assert st == po.OK
            ref = xt[2]
            away = np.abs(np.abs(ref) - np.pi) > 1e-6
            assert away.sum() > T // 2
            assert np.abs(th[b][away] - ref[away]).max() <= TOL * 10
        eng.close()
    finally:
        capi.set_option(capi.OPT_AGENT_LANES, 0)


def test_full_size_batch_properties():
    """BASELINE config 4 size (4096 agents, K = 10, T = 200): size-independent properties.
    (a) mode (0,0) of c_k is exactly the mean of ones; (b) an agent's result does not depend on
    its position in the batch (bitwise); (c) rolling out the updated controls reproduces the
    next call's rollout shifted by one step; (d) controls respect the clamp limits."""
    B = 4096
    eng, _ = make_pair("omni", 10, 20.0, n_oracles=0)
    T, K2 = eng.T, eng.K2
    rng = np.random.default_rng(12345)
    poses = random_poses(rng, B)
    poses[B - 1] = poses[0]
    d_pose = dev(poses)
    d_ut = torch.zeros((B, T, 3), dtype=torch.float64, device="cuda")
    d_u0 = torch.empty((B, 3), dtype=torch.float64, device="cuda")
    d_ck = torch.empty((B, K2), dtype=torch.float64, device="cuda")
    d_traj = torch.empty((B, T, 3), dtype=torch.float64, device="cuda")
    for _ in range(3):
        eng.control_batch(B, d_pose, d_ut, d_u0, ck=d_ck, traj=d_traj)
    torch.cuda.synchronize()
    ck, ut, u0 = d_ck.cpu().numpy(), d_ut.cpu().numpy(), d_u0.cpu().numpy()
    assert np.abs(ck[:, 0] - 1.0).max() < 1e-14
    assert np.array_equal(ut[0], ut[B - 1]) and np.array_equal(u0[0], u0[B - 1])
    assert np.array_equal(u0, ut[:, 0, :])
    lim = np.array([1.0, 1.0, 2.0])
    assert (np.abs(ut) <= lim + 0.0).all()
    # (c): rollout of shifted controls == traj of the next control call
    shifted = np.concatenate([ut[:, 1:, :], np.zeros((B, 1, 3))], axis=1)
    d_sh = dev(shifted)
    d_tr2 = torch.empty((B, T, 3), dtype=torch.float64, device="cuda")
    eng.rollout_batch(B, d_pose, d_sh, d_tr2)
    eng.control_batch(B, d_pose, d_ut, d_u0, traj=d_traj)
    torch.cuda.synchronize()
    assert torch.equal(d_tr2, d_traj)
    # (b) again, against a B = 1 launch of the same agent
    d_ut1 = dev(ut[7:8])
    d_u01 = torch.empty((1, 3), dtype=torch.float64, device="cuda")
    eng.control_batch(1, d_pose[7:8], d_ut1, d_u01)
    torch.cuda.synchronize()
    assert torch.equal(d_ut1[0], d_ut[7])
    eng.close()


def test_ragged_memory_and_extreme_sizes():
    """Edge cases: per-agent memory lengths 0 / 3 / full stride in one batch (d_n_mem), the shortest
    legal horizon (T = 2), a single mode (K = 1) and the largest supported basis (K = 32)."""
    from tests.gpu_util import MODELS
    rng = np.random.default_rng(77)
    # ragged memory: one launch, different n_mem per agent
    eng, ors = make_pair("omni", 10, 5.0, n_oracles=4)
    B, T, K2, stride = 4, eng.T, eng.K2, 100
    n_mem = np.array([0, 3, 100, 41], dtype=np.int32)
    poses = random_poses(rng, B)
    mem = random_poses(rng, B * stride).reshape(B, stride, 3)
    ut0 = rng.uniform(-0.4, 0.4, (B, T, 3))
    d_ut, d_u0 = dev(ut0), torch.empty((B, 3), dtype=torch.float64, device="cuda")
    d_ck = torch.empty((B, K2), dtype=torch.float64, device="cuda")
    eng.control_batch(B, dev(poses), d_ut, d_u0, mem_cols=dev(mem), n_mem=torch.as_tensor(n_mem).cuda(),
                      mem_stride=stride, ck=d_ck)
    torch.cuda.synchronize()
    for b in range(B):
        ors[b].ut = ut0[b].T
        u, st = ors[b].control(MAP_BOUNDS, poses[b], mem[b, :n_mem[b]].T if n_mem[b] else None, stages=True)
        assert np.abs(d_ck[b].cpu().numpy() - st["ck"]).max() < TOL_CK
        assert np.abs(d_u0[b].cpu().numpy() - u).max() < TOL
        assert np.abs(d_ut[b].cpu().numpy().T - st["ut"]).max() < TOL
    eng.close()
    # shortest legal horizon, one mode, largest basis
    run_batch_vs_oracle("simple_cart", 10, 0.2, 0.1, B=2, n_mem=0, calls=2, seed=1)   # T = 2
    run_batch_vs_oracle("omni", 1, 1.0, 0.1, B=2, n_mem=2, calls=2, seed=2)           # K = 1
    run_batch_vs_oracle("omni", 32, 3.0, 0.1, B=1, n_mem=0, calls=1, seed=4)          # K = 32
    # truncating steps_: horizon 0.3 / dt 0.1 -> 2 steps (SURVEY.md section 8 notation)
    eng2, _ = make_pair("omni", 5, 0.3, n_oracles=0)
    assert eng2.T == 2
    eng2.close()


@pytest.mark.parametrize("seed", range(12))
def test_random_shapes_against_oracle(seed):
    """Randomised shapes: basis size 1..32 (exact-K and run-time-K kernels), horizons of 2..420 steps
    (one, two and four wavefronts per agent, chunked scans beyond 256), ragged replay memory,
    both kinematic models; dt = 1/16 keeps horizon / dt exact."""
    rng = np.random.default_rng(1000 + seed)
    model = ["omni", "simple_cart"][seed % 2]
    K = int(rng.choice([1, 2, 3, 5, 6, 9, 10, 11, 13, 16, 17, 20, 24, 30, 32]))
    steps = int(rng.choice([2, 3, 7, 31, 64, 65, 97, 128, 160, 200, 257, 420]))
    n_mem = int(rng.choice([0, 0, 1, 5, 33, 100]))
    dt = 0.0625
    # the oracle costs O(K^2 (T + n_mem)) trig calls per agent and call: bound the batch
    B = 2 if K * K * (steps + n_mem) > 60000 else 4
    run_batch_vs_oracle(model, K, steps * dt, dt, B=B, n_mem=n_mem, calls=2, seed=seed)


@pytest.mark.parametrize("seed", range(12))
def test_random_shapes_fp32_engine(seed):
    """The same randomised shapes on the fp32 engine against the fp64 oracle: controls <= 1e-4,
    co-state <= 5e-4, c_k <= 1e-5 (SURVEY.md 8(d) fp32 bars; measured worst 2.6e-5 / 2.8e-4 / 3e-6)."""
    rng = np.random.default_rng(1000 + seed)
    model = ["omni", "simple_cart"][seed % 2]
    K = int(rng.choice([1, 2, 3, 5, 6, 9, 10, 11, 13, 16, 17, 20, 24, 30, 32]))
    steps = int(rng.choice([2, 3, 7, 31, 64, 65, 97, 128, 160, 200, 257, 420]))
    n_mem = int(rng.choice([0, 0, 1, 5, 33, 100]))
    B = 2 if K * K * (steps + n_mem) > 60000 else 4
    worst = run_batch_vs_oracle(model, K, steps * 0.0625, 0.0625, B=B, n_mem=n_mem, calls=2, seed=seed,
                                precision=capi.PREC_F32, tol=5e-4, tol_ck=1e-5)
    assert worst["u0"] <= 1e-4 and worst["ut"] <= 1e-4, worst


def test_longest_horizon_and_lds_limit():
    """Maximum sizes: a horizon of 1500 steps (K = 10, fp64: 140 KB of the 160 KB LDS, six chunks of
    256 steps with carried scans) still matches the oracle; one that cannot fit is refused with
    EEA_ERR_UNSUPPORTED at launch instead of producing garbage."""
    run_batch_vs_oracle("omni", 10, 1500 * 0.0625, 0.0625, B=2, n_mem=0, calls=1, seed=77)
    eng, _ = make_pair("omni", 10, 2600 * 0.0625, dt=0.0625, n_oracles=0)
    T = eng.T
    assert T == 2600
    d_pose = dev(random_poses(np.random.default_rng(1), 1))
    d_ut = torch.zeros((1, T, 3), dtype=torch.float64, device="cuda")
    d_u0 = torch.empty((1, 3), dtype=torch.float64, device="cuda")
    with pytest.raises(capi.EngineError) as ei:
        eng.control_batch(1, d_pose, d_ut, d_u0)
    assert ei.value.status == capi.ERR_UNSUPPORTED and "LDS" in str(ei.value)
    eng.close()
