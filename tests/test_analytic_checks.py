"""Analytic consequences of the reference's formulas that no transcription can fake, evaluated with mpmath at 40
digits.  They pin the stages the reference's own tests do not reach -- Basis::trajCoeff,
Basis::spatialCoeff, gradErgodicMetric, the backward pass, updateControl -- against closed forms instead of
against a second restatement:

  1. stationary trajectory  =>  c_k = fourierBasis(x)                 (basis.cpp:79-89,109-120)
  2. single-cell target     =>  phi_k = fourierBasis(cell)            (basis.cpp:122-133)
  3. c_k == phi_k           =>  edx == 0, rho == 0, u == clamp(0)     (ergodic_control.hpp:418-451, integrator.hpp:154-194)
  4. (round 4) the closed form of EVERY stage for arbitrary controls, each evaluated from the previous stage's own
     output (tests/analytic_chain.py): RK4 on heading-only kinematics = Simpson's rule, c_k = the mean of the basis,
     edx = q sum lambda_k (c_k - phi_k) grad f_k, the barrier, RK4 on the linear co-state equation with nilpotent A^T =
     its two-term flow, u = clamp(-Rinv B^T rho).  The GPU version of 4 does not touch the oracle at all.

Checked for the oracle (CPU) and for the HIP path (GPU), up to K = 32 -- which also says on which side of the
1e-11 bar the kernels' Chebyshev tables sit at the largest basis.
"""
import mpmath as mp
import numpy as np
import pytest

from oracle import pyoracle as po

mp.mp.dps = 40
BOUNDS = (-1.0, 11.0, -1.0, 5.0)          # lx = 12, ly = 6
MEANS, SIGMAS = [[2.5, 2.5], [8.5, 2.5]], [[1.5, 1.5], [1.5, 1.5]]
LIM = np.array([1.0, 1.0, 2.0])
RINV = np.diag([1.0, 1.0, 2.0])


def basis_mp(K, lx, ly, x, y):
    """f_k(x, y) = cos(k1 pi x / lx) cos(k2 pi y / ly), col = k2 K + k1, from the exact values of the doubles"""
    x, y, lx, ly = mp.mpf(x), mp.mpf(y), mp.mpf(lx), mp.mpf(ly)
    cx = [mp.cos(k * mp.pi * x / lx) for k in range(K)]
    cy = [mp.cos(k * mp.pi * y / ly) for k in range(K)]
    return np.array([float(cx[k1] * cy[k2]) for k2 in range(K) for k1 in range(K)])


def accumulated(n, res):
    """configTarget's coordinates: 0, res, res + res, ... (ergodic_control.hpp:387-408)"""
    out, v = [], 0.0
    for _ in range(n):
        out.append(v)
        v += res
    return out


# ---------------------------------------------------------------------------------------------------------------
# oracle (CPU)
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("K", [5, 10, 20, 32])
def test_oracle_stationary_trajectory_gives_the_basis(K):
    pose = np.array([3.7, 1.9, 0.4])
    o = po.ErgodicControl(po.MODEL_OMNI, 0.1, 2.0, 0.1, 1.0, K, RINV, -LIM, LIM)
    o.set_target(MEANS, SIGMAS)
    _, st = o.control(BOUNDS, pose, stages=True)          # zero warm start: the rollout stays at the pose
    assert np.abs(st["traj"][:2] - pose[:2, None]).max() == 0.0
    ref = basis_mp(K, 12.0, 6.0, pose[0] - BOUNDS[0], pose[1] - BOUNDS[2])
    assert np.abs(st["ck"] - ref).max() < 1e-13


@pytest.mark.parametrize("K,nx,ny,cell", [(10, 121, 61, (17, 40)), (32, 64, 48, (63, 47)), (20, 256, 256, (0, 0))])
def test_oracle_single_cell_target_gives_the_basis_at_the_cell(K, nx, ny, cell):
    res = 0.1
    lx, ly = (nx - 1) * res, (ny - 1) * res
    phi = np.zeros(nx * ny)
    phi[cell[1] * nx + cell[0]] = 1.0
    got = po.spatial_coeff(lx, ly, K, phi, po.phi_grid(nx, ny, res))
    ref = basis_mp(K, lx, ly, accumulated(nx, res)[cell[0]], accumulated(ny, res)[cell[1]])
    assert np.abs(got - ref).max() < 1e-13


@pytest.mark.parametrize("model", [po.MODEL_OMNI, po.MODEL_SIMPLE_CART])
def test_oracle_zero_gradient_gives_zero_costate(model):
    cart = model == po.MODEL_SIMPLE_CART
    lim = np.array([1.0, 0.0, 2.0]) if cart else LIM
    o = po.ErgodicControl(model, 0.1, 5.0, 0.1, 1.0, 10, np.diag([1.0, 0.0, 2.0]) if cart else RINV, -lim, lim)
    o.set_target(MEANS, SIGMAS)
    o.config_target(BOUNDS)
    rng = np.random.default_rng(3)
    ut = rng.uniform(-0.2, 0.2, (3, o.T))
    if cart:
        ut[1] = 0.0
    o.ut = ut
    o.set_shared_ck(o.phik)                                # fourier_diff = lamdak % (phik - phik) = 0
    u, st = o.control(BOUNDS, [5.0, 2.0, 0.3], stages=True)
    assert np.abs(st["bdx"]).max() == 0.0                  # the rollout stays inside the map: no barrier
    assert np.abs(st["edx"]).max() == 0.0 and np.abs(st["rhot"]).max() == 0.0
    assert np.abs(st["ut"]).max() == 0.0 and np.abs(u).max() == 0.0


CHAIN_CASES = [  # model, K, horizon, dt, n_mem, pose (the last two start at / beyond the border: the barrier is active)
    ("omni", 10, 4.0, 0.1, 0, (3.7, 1.9, 0.4)),
    ("simple_cart", 10, 4.0, 0.1, 7, (8.1, 3.3, -2.6)),
    ("omni", 5, 0.5, 0.1, 0, (0.2, 0.7, 3.0)),
    ("omni", 7, 3.0, 0.05, 3, (10.97, 4.99, 1.2)),
    ("simple_cart", 12, 2.5, 0.125, 0, (-1.02, -0.9, 0.7)),
]
CHAIN_MODELS = {"omni": (po.MODEL_OMNI, [1.0, 1.0, 2.0], [1.0, 1.0, 2.0]),
                "simple_cart": (po.MODEL_SIMPLE_CART, [1.0, 0.0, 2.0], [1.0, 0.0, 2.0])}


def _chain_inputs(model, T, n_mem, seed):
    rng = np.random.default_rng(seed)
    ut = rng.uniform(-0.6, 0.6, (3, T))
    ut[2] = rng.uniform(-1.5, 1.5, T)
    if model == "simple_cart":
        ut[1] = 0.0
    shifted = np.concatenate([ut[:, 1:], np.zeros((3, 1))], axis=1)   # ergodic_control.hpp:233-234
    mem = None
    if n_mem:
        mem = np.stack([rng.uniform(-0.5, 10.5, n_mem), rng.uniform(-0.5, 4.5, n_mem), rng.uniform(-3, 3, n_mem)])
    return ut, shifted, mem


def _assert_chain(errs, tol, tol_ck):
    for k, (e, mag) in errs.items():
        bar = (tol_ck if k == "ck" else tol) * max(1.0, mag)
        assert e <= bar, (k, e, mag, errs)


@pytest.mark.parametrize("model,K,horizon,dt,n_mem,pose", CHAIN_CASES)
def test_oracle_every_stage_against_its_closed_form(model, K, horizon, dt, n_mem, pose):
    from tests.analytic_chain import chain_errors
    om, rdiag, lim = CHAIN_MODELS[model]
    lim = np.array(lim)
    o = po.ErgodicControl(om, dt, horizon, 0.1, 1.0, K, np.diag(rdiag), -lim, lim)
    o.set_target(MEANS, SIGMAS)
    o.config_target(BOUNDS)
    ut, shifted, mem = _chain_inputs(model, o.T, n_mem, 17 * K + o.T)
    o.ut = ut
    _, st = o.control(BOUNDS, np.array(pose), mem, stages=True)
    errs = chain_errors(model, K, dt, 1.0, rdiag, -lim, lim, BOUNDS, pose, shifted, st, o.phik, mem)
    # the stages are double-precision sums of up to K^2 T terms: 1e-12 relative to the stage's size (measured <= 4e-14)
    _assert_chain(errs, 1e-12, 1e-14)
    if pose[0] > 10.9 or pose[0] < -1.0:
        assert errs["bdx"][1] > 0.0                                        # the barrier took part ...
        assert (np.abs(st["ut"]) == lim[:, None])[lim > 0].any()           # ... and so did the clamp


def phik_gaussians_mp(K, bounds, res, nx, ny, means, sigmas):
    """configTarget for a Gaussian target as formulas (target.hpp:91-102, target.cpp:78-90, basis.cpp:122-133,
    ergodic_control.hpp:387-411): g(p) = sum_g exp(-1/2 sum_d (p_d - (mu_d - origin_d))^2 / sigma_d^2) on the grid whose
    coordinates are accumulated sums of the resolution, phi = g / sum g, phi_k = sum_cells phi f_k(cell); col = k2 K + k1"""
    lx, ly = mp.mpf(bounds[1]) - mp.mpf(bounds[0]), mp.mpf(bounds[3]) - mp.mpf(bounds[2])
    xs, ys = [mp.mpf(v) for v in accumulated(nx, res)], [mp.mpf(v) for v in accumulated(ny, res)]
    gx = [[mp.exp(-((x - (mp.mpf(m[0]) - mp.mpf(bounds[0]))) ** 2) / (2 * mp.mpf(sg[0]) ** 2)) for x in xs]
          for m, sg in zip(means, sigmas)]
    gy = [[mp.exp(-((y - (mp.mpf(m[1]) - mp.mpf(bounds[2]))) ** 2) / (2 * mp.mpf(sg[1]) ** 2)) for y in ys]
          for m, sg in zip(means, sigmas)]
    G = len(means)
    mass = sum(sum(gx[g]) * sum(gy[g]) for g in range(G))
    cx = [[mp.cos(k * mp.pi * x / lx) for x in xs] for k in range(K)]
    cy = [[mp.cos(k * mp.pi * y / ly) for y in ys] for k in range(K)]
    ax = [[sum(a * b for a, b in zip(gx[g], cx[k])) for k in range(K)] for g in range(G)]   # the sums factor per axis
    ay = [[sum(a * b for a, b in zip(gy[g], cy[k])) for k in range(K)] for g in range(G)]
    return np.array([float(sum(ax[g][k1] * ay[g][k2] for g in range(G)) / mass) for k2 in range(K) for k1 in range(K)])


PHIK_CASES = [  # K, bounds, resolution, means, sigmas
    (6, (0.0, 3.0, 0.0, 2.0), 0.1, [[1.0, 0.7]], [[0.4, 0.3]]),
    (10, BOUNDS, 0.1, MEANS, SIGMAS),
    (12, (-2.0, 4.4, 1.0, 3.55), 0.05, [[0.5, 2.0], [3.9, 3.4], [-1.5, 1.2]], [[0.8, 0.3], [0.2, 0.6], [1.1, 1.1]]),
]


@pytest.mark.parametrize("K,bounds,res,means,sigmas", PHIK_CASES)
def test_oracle_gaussian_target_phik_against_the_formulas(K, bounds, res, means, sigmas):
    o = po.ErgodicControl(po.MODEL_OMNI, 0.1, 1.0, res, 1.0, K, RINV, -LIM, LIM)
    o.set_target(means, sigmas)
    o.config_target(bounds)
    # grid size: round(l / resolution) + 1 (grid.hpp:61-64, ergodic_control.hpp:383-385; pinned by the survey anchors)
    nx, ny = int(round((bounds[1] - bounds[0]) / res)) + 1, int(round((bounds[3] - bounds[2]) / res)) + 1
    ref = phik_gaussians_mp(K, bounds, res, nx, ny, means, sigmas)
    assert abs(ref[0] - 1.0) < 1e-15 and np.abs(o.phik - ref).max() < 1e-14


# ---------------------------------------------------------------------------------------------------------------
# HIP path (through the C ABI)
# ---------------------------------------------------------------------------------------------------------------
def _gpu():
    torch = pytest.importorskip("torch")
    from ergodic_exploration_amd import capi
    return torch, capi


@pytest.mark.gpu
@pytest.mark.parametrize("K,horizon", [(5, 0.5), (10, 20.0), (16, 20.0), (20, 5.0), (30, 6.0), (32, 2.0)])
def test_gpu_stationary_trajectory_gives_the_basis(K, horizon):
    """c_k of a trajectory that never moves against cos(k1 pi x / lx) cos(k2 pi y / ly) at 40 digits: both control
    kernels, every contraction form (4x4 blocks at K = 5 / 10, 16x16 tiles, two tiles per axis, the workgroup kernel
    at K = 30 / 32).  Bar 1e-11; measured margin printed with EEA_PRINT_WORST."""
    import os
    torch, capi = _gpu()
    rng = np.random.default_rng(K)
    B = 7
    poses = np.stack([rng.uniform(0.0, 10.0, B), rng.uniform(0.0, 4.0, B), rng.uniform(-3, 3, B)], 1)
    eng = capi.Engine(capi.make_config(capi.MODEL_OMNI, 0.1, horizon, 0.1, 1.0, K, RINV, -LIM, LIM))
    eng.set_target_gaussians(MEANS, SIGMAS)
    eng.config_domain(BOUNDS)
    d_pose = torch.as_tensor(poses).cuda()
    d_ut = torch.zeros((B, eng.T, 3), dtype=torch.float64, device="cuda")
    d_u0 = torch.empty((B, 3), dtype=torch.float64, device="cuda")
    d_ck = torch.empty((B, K * K), dtype=torch.float64, device="cuda")
    eng.control_batch(B, d_pose, d_ut, d_u0, ck=d_ck)
    torch.cuda.synchronize()
    ck = d_ck.cpu().numpy()
    worst = max(np.abs(ck[b] - basis_mp(K, 12.0, 6.0, poses[b, 0] - BOUNDS[0], poses[b, 1] - BOUNDS[2])).max() for b in range(B))
    if os.environ.get("EEA_PRINT_WORST"):
        print("stationary trajectory: K=%d T=%d  max |c_k - basis(40 digits)| = %.2e" % (K, eng.T, worst))
    assert worst < 1e-11
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("K,nx,ny,cell", [(10, 121, 61, (17, 40)), (32, 64, 48, (63, 47)), (20, 256, 256, (0, 0)),
                                          (30, 1024, 1024, (1000, 513))])
def test_gpu_single_cell_target_gives_the_basis_at_the_cell(K, nx, ny, cell):
    import os
    torch, capi = _gpu()
    res = 0.1
    lx, ly = (nx - 1) * res, (ny - 1) * res
    phi = torch.zeros((nx * ny,), dtype=torch.float64, device="cuda")
    phi[cell[1] * nx + cell[0]] = 1.0
    eng = capi.Engine(capi.make_config(capi.MODEL_OMNI, 0.1, 2.0, res, 1.0, K, RINV, -LIM, LIM))
    eng.set_target_grid(nx, ny, phi, lx, ly)
    got = eng.phik()
    ref = basis_mp(K, lx, ly, accumulated(nx, res)[cell[0]], accumulated(ny, res)[cell[1]])
    worst = np.abs(got - ref).max()
    if os.environ.get("EEA_PRINT_WORST"):
        print("single-cell target: K=%d grid %dx%d  max |phi_k - basis(40 digits)| = %.2e" % (K, nx, ny, worst))
    assert worst < 1e-11
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("model,K,horizon", [("omni", 10, 20.0), ("simple_cart", 10, 20.0), ("omni", 20, 5.0), ("omni", 30, 6.0)])
def test_gpu_zero_gradient_gives_zero_costate(model, K, horizon):
    """the shared c_k set to phi_k makes fourier_diff vanish: inside the map (no barrier) the co-state is exactly
    zero and every control becomes clamp(0) = 0 -- exact zeros, not small numbers"""
    torch, capi = _gpu()
    cart = model == "simple_cart"
    lim = np.array([1.0, 0.0, 2.0]) if cart else LIM
    eng = capi.Engine(capi.make_config(capi.MODEL_SIMPLE_CART if cart else capi.MODEL_OMNI, 0.1, horizon, 0.1, 1.0, K,
                                       np.diag([1.0, 0.0, 2.0]) if cart else RINV, -lim, lim))
    eng.set_target_gaussians(MEANS, SIGMAS)
    eng.config_domain(BOUNDS)
    T, B = eng.T, 5
    rng = np.random.default_rng(3)
    ut = rng.uniform(-0.05, 0.05, (B, T, 3))
    if cart:
        ut[:, :, 1] = 0.0
    poses = np.stack([rng.uniform(4.0, 6.0, B), rng.uniform(1.5, 2.5, B), rng.uniform(-3, 3, B)], 1)
    d_pose, d_ut = torch.as_tensor(poses).cuda(), torch.as_tensor(ut).cuda()
    d_u0 = torch.full((B, 3), 7.0, dtype=torch.float64, device="cuda")
    outs = {k: torch.full((B, T, 3), 7.0, dtype=torch.float64, device="cuda") for k in ("edx", "bdx", "rhot")}
    d_phik = torch.as_tensor(eng.phik()).cuda()
    eng.control_batch(B, d_pose, d_ut, d_u0, ck_shared=d_phik, **outs)
    torch.cuda.synchronize()
    assert float(outs["bdx"].abs().max()) == 0.0
    assert float(outs["edx"].abs().max()) == 0.0 and float(outs["rhot"].abs().max()) == 0.0
    assert float(d_ut.abs().max()) == 0.0 and float(d_u0.abs().max()) == 0.0
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("model,K,horizon,dt,n_mem,pose", CHAIN_CASES + [
    ("omni", 20, 1.0, 0.02, 0, (4.0, 2.0, 0.1)),          # K = 20 wavefront instance (2 x 2 tiles in fp64)
    ("omni", 30, 4.0, 0.1, 5, (6.0, 1.0, -1.0)),          # workgroup-per-agent kernel
    ("simple_cart", 10, 19.6, 0.1, 0, (2.0, 3.0, 2.0)),   # four steps per lane, cooperative last slot (T = 196)
])
def test_gpu_every_stage_against_its_closed_form(model, K, horizon, dt, n_mem, pose, f32=False):
    """The HIP path's stage outputs against the closed forms of tests/analytic_chain.py -- no oracle anywhere in this
    test: engine -> stage outputs -> mpmath.  Bars: the parity bars (1e-9 relative to the stage, c_k 1e-11; the fp32
    engine: 5e-4 / 1e-5, its inputs rounded to float before the closed forms see them)."""
    torch, capi = _gpu()
    from tests.analytic_chain import chain_errors
    _, rdiag, lim = CHAIN_MODELS[model]
    lim = np.array(lim)
    em = capi.MODEL_OMNI if model == "omni" else capi.MODEL_SIMPLE_CART
    tdt = torch.float32 if f32 else torch.float64
    eng = capi.Engine(capi.make_config(em, dt, horizon, 0.1, 1.0, K, np.diag(rdiag), -lim, lim,
                                       precision=capi.PREC_F32 if f32 else capi.PREC_F64))
    eng.set_target_gaussians(MEANS, SIGMAS)
    eng.config_domain(BOUNDS)
    T = eng.T
    ut, shifted, mem = _chain_inputs(model, T, n_mem, 17 * K + T)
    if f32:   # what the engine is given
        r32 = lambda a: None if a is None else np.asarray(a, dtype=np.float32).astype(np.float64)
        ut, shifted, mem, pose = r32(ut), r32(shifted), r32(mem), tuple(r32(np.array(pose)))
    B = 3   # three agents, the same inputs at batch positions 0 .. 2 (the middle one is checked)
    dev = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=tdt).cuda()
    d_pose = dev(np.tile(np.array(pose), (B, 1)))
    d_ut = dev(np.tile(ut.T[None], (B, 1, 1)))
    d_u0 = torch.empty((B, 3), dtype=tdt, device="cuda")
    outs = {k: torch.empty((B, T, 3), dtype=tdt, device="cuda") for k in ("traj", "edx", "bdx", "rhot")}
    d_ck = torch.empty((B, K * K), dtype=tdt, device="cuda")
    d_mem = dev(np.tile(mem.T[None], (B, 1, 1))) if n_mem else None
    d_nmem = torch.full((B,), n_mem, dtype=torch.int32, device="cuda") if n_mem else None
    eng.control_batch(B, d_pose, d_ut, d_u0, mem_cols=d_mem, n_mem=d_nmem, mem_stride=n_mem, ck=d_ck, **outs)
    torch.cuda.synchronize()
    st = {k: v[1].cpu().numpy().astype(np.float64).T for k, v in outs.items()}
    st["ck"] = d_ck[1].cpu().numpy().astype(np.float64)
    st["ut"] = d_ut[1].cpu().numpy().astype(np.float64).T
    errs = chain_errors(model, K, dt, 1.0, rdiag, -lim, lim, BOUNDS, pose, shifted, st, eng.phik(), mem)
    import os
    if os.environ.get("EEA_PRINT_WORST"):
        print("closed-form chain%s" % (" (fp32 engine)" if f32 else ""), model, K, T, n_mem,
              {k: "%.1e (|stage| %.1e)" % v for k, v in errs.items()})
    if f32:
        _assert_chain(errs, 5e-4, 1e-5)
    else:
        _assert_chain(errs, 1e-9, 1e-11)
    assert np.abs(d_u0[1].cpu().numpy().astype(np.float64) - st["ut"][:, 0]).max() == 0.0   # u0 = ut.col(0) (ergodic_control.hpp:310)
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("model,K,horizon,dt,n_mem,pose", [
    ("omni", 20, 5.0, 0.02, 0, (4.0, 2.0, 0.1)),           # BASELINE configs[2]'s instance: outer-product contraction, pair gradient
    ("omni", 20, 2.58, 0.02, 33, (9.0, 3.5, -2.0)),         # T = 129: three steps per lane (the pair's second step dropped), memory
    ("simple_cart", 10, 4.0, 0.1, 7, (8.1, 3.3, -2.6)),
    ("omni", 16, 3.0, 0.1, 0, (2.0, 1.0, 1.0)),
])
def test_gpu_fp32_engine_every_stage_against_its_closed_form(model, K, horizon, dt, n_mem, pose):
    test_gpu_every_stage_against_its_closed_form(model, K, horizon, dt, n_mem, pose, f32=True)


@pytest.mark.gpu
@pytest.mark.parametrize("impl", [0, 1])
@pytest.mark.parametrize("K,bounds,res,means,sigmas", PHIK_CASES + [
    (20, (0.0, 25.5, 0.0, 25.5), 0.1, [[6.0, 6.0], [19.0, 12.0]], [[3.0, 3.0], [3.0, 3.0]]),   # BASELINE configs[2] grid
])
def test_gpu_gaussian_target_phik_against_the_formulas(K, bounds, res, means, sigmas, impl):
    """configTarget's phi_k of a Gaussian target on the HIP path -- both rebuild implementations (per-axis factors in one
    launch; Target::fill + streaming spatialCoeff) -- against the formulas at 40 digits: no oracle in the loop."""
    torch, capi = _gpu()
    capi.set_option(capi.OPT_REBUILD_IMPL, impl)
    try:
        eng = capi.Engine(capi.make_config(capi.MODEL_OMNI, 0.1, 1.0, res, 1.0, K, RINV, -LIM, LIM))
        eng.set_target_gaussians(means, sigmas)
        eng.config_domain(bounds)
        _, nx, ny = eng.target_grid()
        got = eng.phik()
        eng.close()
    finally:
        capi.set_option(capi.OPT_REBUILD_IMPL, 0)
    ref = phik_gaussians_mp(K, bounds, res, nx, ny, means, sigmas)
    err = float(np.abs(got - ref).max())
    import os
    if os.environ.get("EEA_PRINT_WORST"):
        print("closed-form phi_k, rebuild implementation %d, K = %d, %d x %d grid: %.1e" % (impl, K, nx, ny, err))
    assert err < 1e-11
