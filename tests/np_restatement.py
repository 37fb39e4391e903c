"""Second, INDEPENDENT restatement of the control loop's stage formulas in plain numpy.

TEST INFRASTRUCTURE.  Written from the reference's source lines (cited per function, paths relative to the
reference root), NOT from oracle/ergodic_oracle.c: different language, different data layout ((3, T) arrays,
whole-array expressions where the reference loops), different author pass.  Its only job is to catch a
transcription slip in the C oracle that the end-to-end anchor vectors cannot localise: both restatements must
produce the same per-stage outputs (tests/test_oracle_pinning.py, tolerance 1e-12) on the committed fixtures
tests/golden/stages_*.npz.  Summation orders are NOT reproduced here (numpy sums pairwise), hence 1e-12, not
bit equality.
"""
import numpy as np

PI = 3.14159265358979323846  # include/ergodic_exploration/numerics.hpp:59


def normalize_angle_PI(rad):
    """numerics.hpp:78-90"""
    q = np.floor((rad + PI) / (2.0 * PI))
    rad = (rad + PI) - q * 2.0 * PI
    if rad < 0.0:
        rad += 2.0 * PI
    return rad - PI


class Omni:
    """models/omni.hpp:164-215"""

    def f(self, x, u):
        return np.array([u[0] * np.cos(x[2]) - u[1] * np.sin(x[2]),
                         u[0] * np.sin(x[2]) + u[1] * np.cos(x[2]), u[2]])

    def fdx(self, x, u):
        A = np.zeros((3, 3))
        A[0, 2] = -u[0] * np.sin(x[2]) - u[1] * np.cos(x[2])
        A[1, 2] = u[0] * np.cos(x[2]) - u[1] * np.sin(x[2])
        return A

    def fdu(self, x):
        return np.array([[np.cos(x[2]), -np.sin(x[2]), 0.0], [np.sin(x[2]), np.cos(x[2]), 0.0], [0.0, 0.0, 1.0]])


class SimpleCart:
    """models/cart.hpp:152-206"""

    def f(self, x, u):
        if not abs(u[1] - 0.0) < 1.0e-12:
            raise ValueError("Invalid twist y-velocity must be 0.")
        return np.array([u[0] * np.cos(x[2]), u[0] * np.sin(x[2]), u[2]])

    def fdx(self, x, u):
        A = np.zeros((3, 3))
        A[0, 2] = -u[0] * np.sin(x[2])
        A[1, 2] = u[0] * np.cos(x[2])
        return A

    def fdu(self, x):
        B = np.zeros((3, 3))
        B[0, 0] = np.cos(x[2])
        B[1, 0] = np.sin(x[2])
        B[2, 2] = 1.0
        return B


class Basis:
    """src/ergodic_exploration/basis.cpp:48-133"""

    def __init__(self, lx, ly, num_basis):
        self.lx, self.ly, self.K = lx, ly, num_basis
        # k_(0, col) = j (x mode, fastest), k_(1, col) = i  (:58-66)
        self.k1 = np.tile(np.arange(num_basis), num_basis).astype(float)
        self.k2 = np.repeat(np.arange(num_basis), num_basis).astype(float)
        self.lamdak = 1.0 / np.power(1.0 + np.sqrt(self.k1 ** 2 + self.k2 ** 2), 1.5)  # :74

    def fourier_basis(self, pts):
        """:79-89 for every column of pts (2, n) -> (K^2, n)"""
        ax = np.outer(self.k1 * (PI / self.lx), pts[0])
        ay = np.outer(self.k2 * (PI / self.ly), pts[1])
        return np.cos(ax) * np.cos(ay)

    def grad_fourier_basis(self, pt):
        """:91-107 -> (2, K^2)"""
        a, b = self.k1 * (PI / self.lx), self.k2 * (PI / self.ly)
        return np.vstack([-a * np.sin(a * pt[0]) * np.cos(b * pt[1]), -b * np.cos(a * pt[0]) * np.sin(b * pt[1])])

    def traj_coeff(self, xt):
        """:109-120"""
        return (1.0 / xt.shape[1]) * self.fourier_basis(xt[:2]).sum(axis=1)

    def spatial_coeff(self, phi_vals, phi_grid):
        """:122-133"""
        return (self.fourier_basis(phi_grid) * phi_vals[None, :]).sum(axis=1)


def rk4_solve_fwd(model, dt, horizon, x0, ut):
    """include/ergodic_exploration/integrator.hpp:135-152, :176-184"""
    steps = int(abs(horizon / dt))
    x = np.array(x0, dtype=float)
    xt = np.empty((3, steps))
    for i in range(steps):
        u = ut[:, i]
        k1 = model.f(x, u)
        k2 = model.f(x + dt * (0.5 * k1), u)
        k3 = model.f(x + dt * (0.5 * k2), u)
        k4 = model.f(x + dt * k3, u)
        x = x + (dt / 6.0) * (k1 + 2.0 * k2 + 2.0 * k3 + k4)
        x[2] = normalize_angle_PI(x[2])
        xt[:, i] = x
    return xt


def rhodot(rho, gdx, dbar, fdx):
    """include/ergodic_exploration/ergodic_control.hpp:65-69"""
    return -gdx - dbar - fdx.T @ rho


def rk4_solve_bwd(model, dt, horizon, rhoT, xt, ut, edx, bdx):
    """integrator.hpp:154-174, :186-194"""
    steps = int(abs(horizon / dt))
    rho = np.array(rhoT, dtype=float)
    rhot = np.empty((3, steps))
    for i in range(steps - 1, -1, -1):
        A = model.fdx(xt[:, i], ut[:, i])
        g, d = edx[:, i], bdx[:, i]
        k1 = rhodot(rho, g, d, A)
        k2 = rhodot(rho - dt * (0.5 * k1), g, d, A)
        k3 = rhodot(rho - dt * (0.5 * k2), g, d, A)
        k4 = rhodot(rho - dt * k3, g, d, A)
        rho = rho - dt / 6.0 * (k1 + 2.0 * k2 + 2.0 * k3 + k4)
        rhot[:, i] = rho
    return rhot


def grad_ergodic_metric(basis, phik, expl_weight, ck, xt):
    """ergodic_control.hpp:418-436"""
    fourier_diff = basis.lamdak * (ck - phik)
    T = xt.shape[1]
    edx = np.zeros((3, T))
    for i in range(T):
        edx[0:2, i] = basis.grad_fourier_basis(xt[:2, i]) @ fourier_diff
    edx[0:2] *= expl_weight
    return edx


def grad_barrier(lx, ly, xt):
    """ergodic_control.hpp:453-474"""
    weight, eps = 25.0, 0.05
    bdx = np.zeros((3, xt.shape[1]))
    bdx[0] += 2.0 * (xt[0] > lx - eps) * (xt[0] - (lx - eps))
    bdx[1] += 2.0 * (xt[1] > ly - eps) * (xt[1] - (ly - eps))
    bdx[0] += 2.0 * (xt[0] < eps) * (xt[0] - eps)
    bdx[1] += 2.0 * (xt[1] < eps) * (xt[1] - eps)
    bdx[0:2] *= weight
    return bdx


def update_control(model, Rinv, umin, umax, xt, rhot):
    """ergodic_control.hpp:438-451"""
    T = xt.shape[1]
    ut = np.empty((3, T))
    for i in range(T):
        u = -Rinv @ model.fdu(xt[:, i]).T @ rhot[:, i]
        ut[:, i] = np.minimum(np.maximum(u, umin), umax)   # std::clamp component-wise
    return ut


def control_stages(model, dt, horizon, expl_weight, K, Rinv, umin, umax, bounds, phik, x, ut_prev, mem_cols=None):
    """One ErgodicControl::control call (ergodic_control.hpp:224-311) from the given warm-start controls,
    with phi_k given (configTarget is a separate stage).  Returns the dict of stage outputs."""
    xmin, xmax, ymin, ymax = bounds
    lx, ly = xmax - xmin, ymax - ymin
    basis = Basis(lx, ly, K)
    T = int(abs(horizon / dt))
    ut = np.zeros((3, T))
    ut[:, :T - 1] = ut_prev[:, 1:]                      # :233-234
    traj = rk4_solve_fwd(model, dt, horizon, x, ut)     # :237
    xt_total = traj if mem_cols is None or mem_cols.size == 0 else np.hstack([mem_cols, traj])  # buffer.cpp:64-111
    xt_total = xt_total.copy()
    xt_total[0] -= xmin                                 # :243-244
    xt_total[1] -= ymin
    xt = xt_total[:, xt_total.shape[1] - T:]            # :264
    ck = basis.traj_coeff(xt_total)                     # :267
    edx = grad_ergodic_metric(basis, phik, expl_weight, ck, xt)   # :270
    bdx = grad_barrier(lx, ly, xt)                      # :273
    rhot = rk4_solve_bwd(model, dt, horizon, np.zeros(3), xt, ut, edx, bdx)  # :277
    ut_new = update_control(model, Rinv, umin, umax, xt, rhot)    # :305
    return {"traj": traj, "ck": ck, "edx": edx, "bdx": bdx, "rhot": rhot, "ut": ut_new, "u0": ut_new[:, 0].copy()}


def config_target_phik(bounds, resolution, K, means, sigmas):
    """configTarget (ergodic_control.hpp:362-416): grid by accumulation, Target::fill (target.cpp:78-89 with
    Gaussian::operator()(pt, trans), target.hpp:91-102), Basis::spatialCoeff.  Returns (phik, phi_vals, nx, ny)."""
    xmin, xmax, ymin, ymax = bounds
    lx, ly = xmax - xmin, ymax - ymin
    nx = int(round((lx - 0.0) / resolution)) + 1     # axis_length (grid.hpp:61-64) + 1
    ny = int(round((ly - 0.0) / resolution)) + 1
    xs, ys = np.empty(nx), np.empty(ny)
    v = 0.0
    for j in range(nx):
        xs[j] = v
        v += resolution
    v = 0.0
    for i in range(ny):
        ys[i] = v
        v += resolution
    grid = np.vstack([np.tile(xs, ny), np.repeat(ys, nx)])   # col = i * nx + j, x fastest
    trans = np.array([xmin, ymin])
    phi = np.zeros(nx * ny)
    for mu, sg in zip(np.asarray(means, dtype=float), np.asarray(sigmas, dtype=float)):
        cov_inv = np.linalg.inv(np.diag(sg ** 2))
        diff = grid - (mu - trans)[:, None]
        phi += np.exp(-0.5 * np.einsum("ip,ij,jp->p", diff, cov_inv, diff))
    phi /= phi.sum()
    basis = Basis(lx, ly, K)
    phik = np.zeros(K * K)
    step = 8192                                             # bounded scratch: K^2 x 8192 doubles per chunk
    for s in range(0, nx * ny, step):
        phik += basis.spatial_coeff(phi[s:s + step], grid[:, s:s + step])
    return phik, phi, nx, ny
