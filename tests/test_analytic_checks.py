"""Analytic consequences of the reference's formulas that no transcription can fake, evaluated with mpmath at 40
digits.  They pin the stages the reference's own tests do not reach -- Basis::trajCoeff,
Basis::spatialCoeff, gradErgodicMetric, the backward pass, updateControl -- against closed forms instead of
against a second restatement:

  1. stationary trajectory  =>  c_k = fourierBasis(x)                 (basis.cpp:79-89,109-120)
  2. single-cell target     =>  phi_k = fourierBasis(cell)            (basis.cpp:122-133)
  3. c_k == phi_k           =>  edx == 0, rho == 0, u == clamp(0)     (ergodic_control.hpp:418-451, integrator.hpp:154-194)

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
