"""Closed forms of EVERY stage of ErgodicControl::control, evaluated with mpmath at 40 digits from the previous stage's
output of the system under test (the oracle, or the HIP path): a chain of independent pins, none of which restates the
reference's loops.

  stage    closed form                                                                      reference
  traj     RK4 on kinematics whose right-hand side depends on the heading only IS Simpson's   integrator.hpp:130-146,176-184
           rule in the heading: x += dt/6 (f(th) + 4 f(th + dt w/2) + f(th + dt w)), th += dt w  models/omni.hpp, cart.hpp:165-173
  ck       (1/N) sum over the trajectory (and the sampled past states) of cos cos            basis.cpp:79-89,109-120
  edx      q sum_k lambda_k (c_k - phi_k) grad f_k(x_i),  lambda_k = (1 + |k|)^-3/2          basis.cpp:71-73,91-107; ergodic_control.hpp:418-436
  bdx      50 ((x > l - eps)(x - (l - eps)) + (x < eps)(x - eps)), eps = 0.05                ergodic_control.hpp:453-474
  rhot     RK4 on the LINEAR co-state equation is the degree-4 Taylor polynomial of its flow;  integrator.hpp:148-194
           A = D1 f has the heading column only, so (A^T)^2 = 0 and the polynomial ends after the linear term:
             rho_xy += dt g_xy,   rho_th += dt (a . rho_xy) + dt^2/2 (a . g_xy),   g = edx_i + bdx_i,  a = A(0:2, 2)(x_i, u_i)
  ut       clamp(-Rinv B(th_i)^T rho_i)                                                     ergodic_control.hpp:438-451

Every function takes numpy doubles (exactly representable inputs) and returns the largest absolute difference between the
system's stage and the closed form, together with the stage's magnitude.
"""
import mpmath as mp
import numpy as np

mp.mp.dps = 40
M = mp.mpf


def _cart(model_name):
    return model_name == "simple_cart"


def _wrap_diff(a, b):
    d = (a - b) / (2 * mp.pi)
    return abs((d - mp.nint(d)) * 2 * mp.pi)


def chain_errors(model_name, K, dt, q, rinv_diag, umin, umax, bounds, pose, u, st, phik, mem=None):
    """u: (3, T) the controls the rollout uses (already shifted: column i drives step i); st: dict with traj, ck, edx, bdx,
    rhot, ut as (3, T) / (K^2,) arrays; phik (K^2,); mem: (3, n_mem) sampled past states or None.
    Returns {stage: (max abs difference, max abs of the closed form)}."""
    T = u.shape[1]
    h = M(dt)
    x0, y0 = M(bounds[0]), M(bounds[2])
    lx, ly = M(bounds[1]) - M(bounds[0]), M(bounds[3]) - M(bounds[2])
    out = {}

    # ---- traj: Simpson's rule in the heading, from the pose and the controls -------------------------------------
    X, Y, TH = M(float(pose[0])), M(float(pose[1])), M(float(pose[2]))
    e_xy, e_th, m_xy = M(0), M(0), M(0)
    for i in range(T):
        vx, vy, w = M(float(u[0, i])), M(float(u[1, i])), M(float(u[2, i]))
        if _cart(model_name):
            vy = M(0)
        sx = sy = M(0)
        for wt, t in ((1, TH), (4, TH + h * w / 2), (1, TH + h * w)):
            c, s = mp.cos(t), mp.sin(t)
            sx += wt * (vx * c - vy * s)
            sy += wt * (vx * s + vy * c)
        X, Y, TH = X + h / 6 * sx, Y + h / 6 * sy, TH + h * w
        e_xy = max(e_xy, abs(M(float(st["traj"][0, i])) - X), abs(M(float(st["traj"][1, i])) - Y))
        e_th = max(e_th, _wrap_diff(M(float(st["traj"][2, i])), TH))
        m_xy = max(m_xy, abs(X), abs(Y))
    out["traj_xy"] = (float(e_xy), float(m_xy))
    out["traj_th"] = (float(e_th), 1.0)

    # ---- tables of the system's OWN trajectory (fourier frame) ---------------------------------------------------
    tx = [M(float(v)) - x0 for v in st["traj"][0]]
    ty = [M(float(v)) - y0 for v in st["traj"][1]]
    th = [M(float(v)) for v in st["traj"][2]]
    ax = [[k * mp.pi * x / lx for k in range(K)] for x in tx]
    ay = [[k * mp.pi * y / ly for k in range(K)] for y in ty]
    cx = [[mp.cos(a) for a in row] for row in ax]
    sx_ = [[mp.sin(a) for a in row] for row in ax]
    cy = [[mp.cos(a) for a in row] for row in ay]
    sy_ = [[mp.sin(a) for a in row] for row in ay]

    # ---- ck: the mean of the basis over the trajectory and the sampled past states -------------------------------
    n_mem = 0 if mem is None else mem.shape[1]
    mcx = [[mp.cos(k * mp.pi * (M(float(mem[0, j])) - x0) / lx) for k in range(K)] for j in range(n_mem)]
    mcy = [[mp.cos(k * mp.pi * (M(float(mem[1, j])) - y0) / ly) for k in range(K)] for j in range(n_mem)]
    N = T + n_mem
    e, mag = M(0), M(0)
    for k2 in range(K):
        for k1 in range(K):
            c = (sum(cx[i][k1] * cy[i][k2] for i in range(T)) + sum(mcx[j][k1] * mcy[j][k2] for j in range(n_mem))) / N
            e = max(e, abs(M(float(st["ck"][k2 * K + k1])) - c))
            mag = max(mag, abs(c))
    out["ck"] = (float(e), float(mag))

    # ---- edx: from the system's own c_k ---------------------------------------------------------------------------
    D = [[(1 / (1 + mp.sqrt(M(k1 * k1 + k2 * k2))) ** M("1.5")) *
          (M(float(st["ck"][k2 * K + k1])) - M(float(phik[k2 * K + k1]))) for k1 in range(K)] for k2 in range(K)]
    e, mag = M(0), M(0)
    for i in range(T):
        gx = -sum(D[k2][k1] * (k1 * mp.pi / lx) * sx_[i][k1] * cy[i][k2] for k2 in range(K) for k1 in range(K)) * M(q)
        gy = -sum(D[k2][k1] * (k2 * mp.pi / ly) * cx[i][k1] * sy_[i][k2] for k2 in range(K) for k1 in range(K)) * M(q)
        e = max(e, abs(M(float(st["edx"][0, i])) - gx), abs(M(float(st["edx"][1, i])) - gy), abs(M(float(st["edx"][2, i]))))
        mag = max(mag, abs(gx), abs(gy))
    out["edx"] = (float(e), float(mag))

    # ---- bdx ----------------------------------------------------------------------------------------------------------
    eps, e, mag = M("0.05"), M(0), M(0)
    for i in range(T):
        for r, (v, l) in enumerate(((tx[i], lx), (ty[i], ly))):
            # eps and l - eps as the doubles the reference forms (0.05 and lx - 0.05 in double)
            epsd = M(0.05)
            hi = M(float(l) - 0.05)
            b = 50 * ((v - hi) if v > hi else M(0)) + 50 * ((v - epsd) if v < epsd else M(0))
            e = max(e, abs(M(float(st["bdx"][r, i])) - b))
            mag = max(mag, abs(b))
        e = max(e, abs(M(float(st["bdx"][2, i]))))
    out["bdx"] = (float(e), float(mag))
    del eps

    # ---- rhot: the nilpotent closed form, from the system's own edx / bdx / headings ----------------------------------
    rx = ry = rt = M(0)
    e, mag = M(0), M(0)
    for i in range(T - 1, -1, -1):
        gx = M(float(st["edx"][0, i])) + M(float(st["bdx"][0, i]))
        gy = M(float(st["edx"][1, i])) + M(float(st["bdx"][1, i]))
        vx, vy = M(float(u[0, i])), (M(0) if _cart(model_name) else M(float(u[1, i])))
        c, s = mp.cos(th[i]), mp.sin(th[i])
        a0, a1 = -vx * s - vy * c, vx * c - vy * s
        rt = rt + h * (a0 * rx + a1 * ry) + h * h / 2 * (a0 * gx + a1 * gy)   # (the old rho_xy on the right-hand side)
        rx, ry = rx + h * gx, ry + h * gy
        e = max(e, abs(M(float(st["rhot"][0, i])) - rx), abs(M(float(st["rhot"][1, i])) - ry),
                abs(M(float(st["rhot"][2, i])) - rt))
        mag = max(mag, abs(rx), abs(ry), abs(rt))
    out["rhot"] = (float(e), float(mag))

    # ---- ut: from the system's own co-state ----------------------------------------------------------------------------
    e, mag = M(0), M(0)
    for i in range(T):
        c, s = mp.cos(th[i]), mp.sin(th[i])
        r0, r1, r2 = (M(float(st["rhot"][r, i])) for r in range(3))
        if _cart(model_name):
            bt = (c * r0 + s * r1, M(0), r2)            # B = [[c,0,0],[s,0,0],[0,0,1]]
        else:
            bt = (c * r0 + s * r1, -s * r0 + c * r1, r2)  # B = [[c,-s,0],[s,c,0],[0,0,1]]
        for r in range(3):
            v = -M(float(rinv_diag[r])) * bt[r]
            v = min(max(v, M(float(umin[r]))), M(float(umax[r])))
            e = max(e, abs(M(float(st["ut"][r, i])) - v))
            mag = max(mag, abs(v))
    out["ut"] = (float(e), float(mag))
    return out
