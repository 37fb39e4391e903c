"""ctypes binding of the CPU oracle (oracle/ergodic_oracle.{h,c}).

TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this module (see ergodic_oracle.h).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libergodic_oracle.so")

MODEL_OMNI, MODEL_SIMPLE_CART, MODEL_CART, MODEL_MECANUM = 0, 1, 2, 3
OK, ERR_INVALID_ARGUMENT, ERR_INVALID_TWIST = 0, 1, 2

_dp = C.POINTER(C.c_double)


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "ergodic_oracle.c")
    hdr = os.path.join(_HERE, "ergodic_oracle.h")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(src), os.path.getmtime(hdr))):
        return _LIB_PATH
    subprocess.check_call(["make", "-s", "-C", _HERE, "libergodic_oracle.so"])
    return _LIB_PATH


class ControlConfig(C.Structure):
    _fields_ = [("model", C.c_int), ("dt", C.c_double), ("horizon", C.c_double),
                ("resolution", C.c_double), ("expl_weight", C.c_double),
                ("num_basis", C.c_uint), ("Rinv", C.c_double * 9),
                ("umin", C.c_double * 3), ("umax", C.c_double * 3)]


class StageOut(C.Structure):
    _fields_ = [(n, _dp) for n in ("traj", "ck", "edx", "bdx", "rhot", "ut")]


class Grid(C.Structure):
    _fields_ = [("xsize", C.c_uint), ("ysize", C.c_uint), ("resolution", C.c_double),
                ("xmin", C.c_double), ("ymin", C.c_double), ("xmax", C.c_double),
                ("ymax", C.c_double), ("data", C.POINTER(C.c_int8))]


class Collision(C.Structure):
    _fields_ = [("boundary_radius", C.c_double), ("search_radius", C.c_double),
                ("obstacle_threshold", C.c_double), ("occupied_threshold", C.c_double)]


class Dwa(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("dt", "horizon", "acc_dt", "acc_lim_x", "acc_lim_y", "acc_lim_th",
                                          "max_vel_x", "min_vel_x", "max_vel_y", "min_vel_y",
                                          "max_rot_vel", "min_rot_vel")] + \
               [(n, C.c_uint) for n in ("vx_samples", "vy_samples", "vth_samples")]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.eo_normalize_angle_PI.restype = C.c_double
        L.eo_normalize_angle_PI.argtypes = [C.c_double]
        L.eo_entropy.restype = C.c_double
        L.eo_entropy.argtypes = [C.c_double]
        L.eo_axis_upper.restype = C.c_double
        L.eo_axis_upper.argtypes = [C.c_double, C.c_double, C.c_uint]
        L.eo_axis_length.restype = C.c_uint
        L.eo_axis_length.argtypes = [C.c_double] * 3
        L.eo_steps.restype = C.c_uint
        L.eo_steps.argtypes = [C.c_double, C.c_double]
        L.eo_target_evaluate.restype = C.c_double
        L.eo_control_phik.restype = _dp
        L.eo_control_lamdak.restype = _dp
        L.eo_control_ut.restype = _dp
        L.eo_control_steps.restype = C.c_uint
        L.eo_bench_control.restype = C.c_double
        L.eo_batch_control.restype = C.c_double
        L.eo_dwa_objective_traj.restype = C.c_double
        L.eo_grid2rowmajor.restype = C.c_uint
        _lib = L
    return _lib


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a):
    return a.ctypes.data_as(_dp)


def normalize_angle_PI(rad):
    return lib().eo_normalize_angle_PI(float(rad))


def steps(horizon, dt):
    return int(lib().eo_steps(float(horizon), float(dt)))


def integrate_twist(x, u, dt):
    x, u = _d(x), _d(u)
    out = np.empty(3)
    lib().eo_integrate_twist(_p(x), _p(u), C.c_double(dt), _p(out))
    return out


def _mp(mp):
    return _d(mp if mp is not None else [0.0, 0.0, 0.0])


def model_f(model, x, u, mp=None):
    x, u, mp = _d(x), _d(u), _mp(mp)
    out = np.empty(3)
    st = lib().eo_model_f(model, _p(mp), _p(x), _p(u), _p(out))
    return st, out


def model_fdx(model, x, u, mp=None):
    x, u, mp = _d(x), _d(u), _mp(mp)
    A = np.empty(9)
    st = lib().eo_model_fdx(model, _p(mp), _p(x), _p(u), _p(A))
    return st, A.reshape(3, 3, order="F")


def model_fdu(model, x, mp=None):
    x, mp = _d(x), _mp(mp)
    nu = lib().eo_model_nu(model)
    B = np.empty(3 * nu)
    st = lib().eo_model_fdu(model, _p(mp), _p(x), _p(B))
    return st, B.reshape(3, nu, order="F")


def model_wheels2twist(model, u, mp):
    u, mp = _d(u), _mp(mp)
    vb = np.empty(3)
    st = lib().eo_model_wheels2twist(model, _p(mp), _p(u), _p(vb))
    return st, vb


def rk4_solve_fwd(model, dt, horizon, x0, ut, mp=None):
    """ut: (n_u, T) array. returns (status, xt (3, T))"""
    x0, mp = _d(x0), _mp(mp)
    utf = _d(np.asarray(ut).T)  # column-major n_u x T == row-major T x n_u
    T = steps(horizon, dt)
    xt = np.empty((T, 3))
    st = lib().eo_rk4_solve_fwd(model, _p(mp), C.c_double(dt), C.c_double(horizon), _p(x0),
                                _p(utf), _p(xt))
    return st, xt.T.copy()


def rk4_solve_bwd(model, dt, horizon, rhoT, xt, ut, edx, bdx, mp=None):
    rhoT, mp = _d(rhoT), _mp(mp)
    a = [_d(np.asarray(m).T) for m in (xt, ut, edx, bdx)]
    T = steps(horizon, dt)
    rhot = np.empty((T, 3))
    st = lib().eo_rk4_solve_bwd(model, _p(mp), C.c_double(dt), C.c_double(horizon), _p(rhoT),
                                _p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), _p(rhot))
    return st, rhot.T.copy()


def basis_init(K):
    k = np.empty((K * K, 2), dtype=np.int64)
    lam = np.empty(K * K)
    lib().eo_basis_init(C.c_uint(K), k.ctypes.data_as(C.POINTER(C.c_int64)), _p(lam))
    return k.T.copy(), lam


def fourier_basis(lx, ly, K, x):
    x = _d(x)
    fk = np.empty(K * K)
    lib().eo_fourier_basis(C.c_double(lx), C.c_double(ly), C.c_uint(K), _p(x), _p(fk))
    return fk


def grad_fourier_basis(lx, ly, K, x):
    x = _d(x)
    dfk = np.empty((K * K, 2))
    lib().eo_grad_fourier_basis(C.c_double(lx), C.c_double(ly), C.c_uint(K), _p(x), _p(dfk))
    return dfk.T.copy()


def traj_coeff(lx, ly, K, xt):
    """xt: (rows>=2, N)"""
    xt = np.asarray(xt)
    rows, n = xt.shape
    xtf = _d(xt.T)
    ck = np.empty(K * K)
    lib().eo_traj_coeff(C.c_double(lx), C.c_double(ly), C.c_uint(K), _p(xtf), C.c_uint(rows),
                        C.c_uint(n), _p(ck))
    return ck


def phi_grid(nx, ny, resolution):
    """Grid of ergodic_control.hpp:387-408 (coordinates by accumulation) as (2, nx*ny)."""
    xs = np.empty(nx)
    ys = np.empty(ny)
    v = 0.0
    for j in range(nx):
        xs[j] = v
        v += resolution
    v = 0.0
    for i in range(ny):
        ys[i] = v
        v += resolution
    g = np.empty((2, nx * ny))
    g[0] = np.tile(xs, ny)
    g[1] = np.repeat(ys, nx)
    return g


def spatial_coeff(lx, ly, K, phi_vals, grid):
    phi_vals = _d(phi_vals)
    gf = _d(np.asarray(grid).T)
    P = phi_vals.shape[0]
    out = np.empty(K * K)
    lib().eo_spatial_coeff(C.c_double(lx), C.c_double(ly), C.c_uint(K), _p(phi_vals), _p(gf),
                           C.c_uint(P), _p(out))
    return out


def target_fill(mu, sigma, trans, grid):
    mu, sigma, trans = _d(mu).reshape(-1), _d(sigma).reshape(-1), _d(trans)
    gf = _d(np.asarray(grid).T)
    P = gf.shape[0]
    out = np.empty(P)
    lib().eo_target_fill(C.c_uint(mu.size // 2), _p(mu), _p(sigma), _p(trans), _p(gf),
                         C.c_uint(P), _p(out))
    return out


class GridMap:
    def __init__(self, xmin, xmax, ymin, ymax, resolution, data):
        self.data = np.ascontiguousarray(data, dtype=np.int8).reshape(-1)
        self.g = Grid()
        st = lib().eo_grid_init(C.byref(self.g), C.c_double(xmin), C.c_double(xmax),
                                C.c_double(ymin), C.c_double(ymax), C.c_double(resolution),
                                self.data.ctypes.data_as(C.POINTER(C.c_int8)),
                                C.c_uint(self.data.size))
        if st != OK:
            raise ValueError("Grid data size does not match the grid size")

    xsize = property(lambda s: s.g.xsize)
    ysize = property(lambda s: s.g.ysize)

    def world2grid(self, x, y):
        ij = (C.c_uint * 2)()
        lib().eo_world2grid(C.byref(self.g), C.c_double(x), C.c_double(y), ij)
        return int(ij[0]), int(ij[1])

    def grid2world(self, i, j):
        xy = (C.c_double * 2)()
        lib().eo_grid2world(C.byref(self.g), C.c_uint(i), C.c_uint(j), xy)
        return float(xy[0]), float(xy[1])

    def grid2rowmajor(self, i, j):
        return int(lib().eo_grid2rowmajor(C.byref(self.g), C.c_uint(i), C.c_uint(j)))

    def rowmajor2grid(self, idx):
        ij = (C.c_uint * 2)()
        lib().eo_rowmajor2grid(C.byref(self.g), C.c_uint(idx), ij)
        return int(ij[0]), int(ij[1])

    def bounds_ij(self, i, j):
        return bool(lib().eo_grid_bounds_ij(C.byref(self.g), C.c_uint(i & 0xFFFFFFFF),
                                            C.c_uint(j & 0xFFFFFFFF)))

    def bounds_idx(self, idx):
        return bool(lib().eo_grid_bounds_idx(C.byref(self.g), C.c_uint(idx & 0xFFFFFFFF)))

    def get_cell(self, idx):
        v = C.c_double()
        st = lib().eo_grid_get_cell(C.byref(self.g), C.c_uint(idx), C.byref(v))
        if st != OK:
            raise ValueError("Grid index out of range")
        return v.value


def collision_check(coll, grid, pose):
    """coll: (boundary_radius, search_radius, obstacle_threshold, occupied_threshold).
    returns (hit, sqrd_obs, dx, dy)"""
    c = Collision(*[float(v) for v in coll])
    pose = _d(pose)
    so, dx, dy = C.c_int(), C.c_int(), C.c_int()
    hit = lib().eo_collision_check(C.byref(c), C.byref(grid.g), _p(pose), C.byref(so),
                                   C.byref(dx), C.byref(dy))
    return bool(hit), so.value, dx.value, dy.value


def validate_control(coll, grid, x0, u, dt, horizon):
    c = Collision(*[float(v) for v in coll])
    x0, u = _d(x0), _d(u)
    return bool(lib().eo_validate_control(C.byref(c), C.byref(grid.g), _p(x0), _p(u),
                                          C.c_double(dt), C.c_double(horizon)))


def dwa_control(dwa, coll, grid, x0, vb, vref=None, xt_ref=None, dt_ref=0.0):
    """DynamicWindow::control; dwa: tuple in Dwa field order. xt_ref (3, n). returns (found, u_opt, cost)"""
    d = Dwa(*dwa)
    c = Collision(*[float(v) for v in coll])
    x0, vb = _d(x0), _d(vb)
    u = np.empty(3)
    cost = C.c_double()
    if xt_ref is None:
        vref = _d(vref)
        ok = lib().eo_dwa_control_vref(C.byref(d), C.byref(c), C.byref(grid.g), _p(x0), _p(vb), _p(vref), _p(u),
                                       C.byref(cost))
    else:
        xr = _d(np.asarray(xt_ref).T)
        ok = lib().eo_dwa_control_traj(C.byref(d), C.byref(c), C.byref(grid.g), _p(x0), _p(vb), _p(xr),
                                       C.c_uint(xr.shape[0]), C.c_double(dt_ref), _p(u), C.byref(cost))
    return bool(ok), u, cost.value


def dwa_objective_traj(dwa, coll, grid, x0, u, xt_ref, dt_ref):
    """trajectory-distance cost of one candidate twist (dynamic_window.cpp:258-286)"""
    d = Dwa(*dwa)
    c = Collision(*[float(v) for v in coll])
    x0, u = _d(x0), _d(u)
    xr = _d(np.asarray(xt_ref).T)
    return float(lib().eo_dwa_objective_traj(C.byref(d), C.byref(c), C.byref(grid.g), _p(x0), _p(u), _p(xr),
                                             C.c_uint(xr.shape[0]), C.c_double(dt_ref)))


def make_config(model, dt, horizon, resolution, expl_weight, num_basis, Rinv, umin, umax):
    cfg = ControlConfig()
    cfg.model = model
    cfg.dt, cfg.horizon, cfg.resolution, cfg.expl_weight = dt, horizon, resolution, expl_weight
    cfg.num_basis = num_basis
    R = np.asarray(Rinv, dtype=np.float64).reshape(3, 3)
    for c in range(3):
        for r in range(3):
            cfg.Rinv[r + 3 * c] = R[r, c]
    for i in range(3):
        cfg.umin[i] = umin[i]
        cfg.umax[i] = umax[i]
    return cfg


class ErgodicControl:
    """Mirror of the reference class for one agent (ergodic_control.hpp:72-185)."""

    def __init__(self, model, dt, horizon, resolution, expl_weight, num_basis, Rinv, umin, umax):
        self.cfg = make_config(model, dt, horizon, resolution, expl_weight, num_basis, Rinv,
                               umin, umax)
        self.h = C.c_void_p()
        st = lib().eo_control_create(C.byref(self.cfg), C.byref(self.h))
        if st != OK:
            raise ValueError("Need at least two steps in forward simulation")
        self.T = int(lib().eo_control_steps(self.h))
        self.K = num_basis

    def __del__(self):
        if getattr(self, "h", None):
            lib().eo_control_destroy(self.h)
            self.h = None

    def set_target(self, mu, sigma):
        mu, sigma = _d(mu).reshape(-1), _d(sigma).reshape(-1)
        lib().eo_control_set_target(self.h, C.c_uint(mu.size // 2), _p(mu), _p(sigma))

    def set_target_grid(self, nx, ny, phi_vals, lx, ly):
        phi_vals = _d(phi_vals)
        lib().eo_control_set_target_grid(self.h, C.c_uint(nx), C.c_uint(ny), _p(phi_vals),
                                         C.c_double(lx), C.c_double(ly))

    def set_shared_ck(self, ck_shared):
        """consensus switch: the shared c_k (K^2) replaces the agent's own in the gradient; None resets"""
        if ck_shared is None:
            lib().eo_control_set_shared_ck(self.h, None)
        else:
            a = _d(ck_shared).reshape(-1)
            assert a.size == self.K * self.K
            lib().eo_control_set_shared_ck(self.h, _p(a))

    def config_target(self, bounds):
        return int(lib().eo_control_config_target(self.h, *[C.c_double(b) for b in bounds]))

    @property
    def phik(self):
        return np.ctypeslib.as_array(lib().eo_control_phik(self.h), (self.K * self.K,)).copy()

    @property
    def lamdak(self):
        return np.ctypeslib.as_array(lib().eo_control_lamdak(self.h), (self.K * self.K,)).copy()

    @property
    def ut(self):
        """(3, T) copy of the warm-start controls"""
        return np.ctypeslib.as_array(lib().eo_control_ut(self.h), (self.T, 3)).T.copy()

    @ut.setter
    def ut(self, v):
        a = np.ctypeslib.as_array(lib().eo_control_ut(self.h), (self.T, 3))
        a[:] = np.asarray(v, dtype=np.float64).T

    def control(self, bounds, x, mem_cols=None, stages=False):
        """bounds = (xmin, xmax, ymin, ymax); mem_cols (3, n_mem) map-frame columns.
        returns u (3,) or (u, dict of (3,T)/(K2,) arrays) when stages=True"""
        x = _d(x)
        if mem_cols is None or np.asarray(mem_cols).size == 0:
            mem, n_mem = None, 0
        else:
            mem = _d(np.asarray(mem_cols).T)
            n_mem = mem.shape[0]
        u = np.empty(3)
        so, bufs = None, {}
        if stages:
            so = StageOut()
            for name in ("traj", "edx", "bdx", "rhot", "ut"):
                bufs[name] = np.empty((self.T, 3))
                setattr(so, name, _p(bufs[name]))
            bufs["ck"] = np.empty(self.K * self.K)
            so.ck = _p(bufs["ck"])
        st = lib().eo_control_step(self.h, *[C.c_double(b) for b in bounds], _p(x),
                                   _p(mem) if mem is not None else None, C.c_uint(n_mem), _p(u),
                                   C.byref(so) if so is not None else None)
        if st == ERR_INVALID_TWIST:
            raise ValueError("Invalid twist y-velocity must be 0.")
        if st != OK:
            raise ValueError("oracle control failed: %d" % st)
        if stages:
            out = {k: (v.T.copy() if v.ndim == 2 else v) for k, v in bufs.items()}
            return u, out
        return u

    def opt_traj(self):
        traj = np.empty((self.T, 3))
        st = lib().eo_control_opt_traj(self.h, _p(traj))
        if st != OK:
            raise ValueError("oracle opt_traj failed: %d" % st)
        return traj.T.copy()


def rk4_step_fwd(model, dt, x, u, mp=None):
    x, u, mp = _d(x), _d(u), _mp(mp)
    out = np.empty(3)
    st = lib().eo_rk4_step_fwd(model, _p(mp), C.c_double(dt), _p(x), _p(u), _p(out))
    return st, out


def bench_control(cfg, mu, sigma, bounds, poses, calls, threads):
    """Timed CPU-baseline loop (bench.py). poses (n_agents, 3). returns (seconds, u_last)"""
    mu, sigma = _d(mu).reshape(-1), _d(sigma).reshape(-1)
    poses = _d(poses)
    n = poses.shape[0]
    u_last = np.empty((n, 3))
    sec = lib().eo_bench_control(C.byref(cfg), C.c_uint(mu.size // 2), _p(mu), _p(sigma),
                                 *[C.c_double(b) for b in bounds], _p(poses), C.c_uint(n),
                                 C.c_uint(calls), C.c_uint(threads), _p(u_last))
    return sec, u_last


def batch_control(cfg, mu, sigma, bounds, poses, calls, threads):
    """One warm-up + `calls` control() calls on every agent (independent controllers, zero warm start,
    one agent per thread).  returns (u_last (n, 3), ut_last (n, T, 3)) after the last call."""
    mu, sigma = _d(mu).reshape(-1), _d(sigma).reshape(-1)
    poses = _d(poses)
    n = poses.shape[0]
    T = steps(cfg.horizon, cfg.dt)
    u_last = np.empty((n, 3))
    ut_last = np.empty((n, T, 3))
    lib().eo_batch_control(C.byref(cfg), C.c_uint(mu.size // 2), _p(mu), _p(sigma),
                           *[C.c_double(b) for b in bounds], _p(poses), C.c_uint(n),
                           C.c_uint(calls), C.c_uint(threads), _p(u_last), _p(ut_last))
    return u_last, ut_last
