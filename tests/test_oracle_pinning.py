"""Pins the CPU oracle (oracle/) against everything the reference offers for this path:
(1) the reference's own 22 gtest known-answer tests, transcribed as data in
    tests/golden/reference_kats.json, and
(2) the end-to-end outputs of the reference's sources recorded in SURVEY.md 8(c)
    (tests/golden/survey_anchors.json).
CPU only.  The oracle is test infrastructure; nothing here touches the product path."""
import math
import os

import numpy as np
import pytest

from oracle import pyoracle as po


def ulp_close(a, b, n=4):
    if a == b:
        return True
    return abs(a - b) <= n * np.spacing(max(abs(a), abs(b)))


# ---- test/test_cart.cpp -----------------------------------------------------------------
def test_cart_kats(kats):
    k = kats["cart"]
    tol = k["tol"]
    st, f = po.model_f(po.MODEL_CART, k["x"], k["u"], k["mp"])
    assert st == po.OK
    np.testing.assert_allclose(f, k["f"], atol=tol, rtol=0)
    st, A = po.model_fdx(po.MODEL_CART, k["x"], k["u"], k["mp"])
    assert abs(A[0, 2] - k["fdx_02"]) < tol and abs(A[1, 2] - k["fdx_12"]) < tol
    st, B = po.model_fdu(po.MODEL_CART, k["x"], k["mp"])
    np.testing.assert_allclose(B, np.array(k["fdu"]), atol=tol, rtol=0)
    for w in k["wheels2twist"]:
        st, vb = po.model_wheels2twist(po.MODEL_CART, w["u"], k["mp"])
        np.testing.assert_allclose(vb, w["vb"], atol=tol, rtol=0)


def test_simple_cart_kats(kats):
    k = kats["simple_cart"]
    tol = k["tol"]
    st, f = po.model_f(po.MODEL_SIMPLE_CART, k["x"], k["u"])
    assert st == po.OK
    np.testing.assert_allclose(f, k["f"], atol=tol, rtol=0)
    st, A = po.model_fdx(po.MODEL_SIMPLE_CART, k["x"], k["u"])
    assert abs(A[0, 2] - k["fdx_02"]) < tol and abs(A[1, 2] - k["fdx_12"]) < tol
    st, B = po.model_fdu(po.MODEL_SIMPLE_CART, k["x"])
    assert abs(B[0, 0] - k["fdu_00"]) < tol and abs(B[1, 0] - k["fdu_10"]) < tol


def test_simple_cart_rejects_lateral_velocity():
    # models/cart.hpp:167-170 throws std::invalid_argument when |u1| >= 1e-12
    st, _ = po.model_f(po.MODEL_SIMPLE_CART, [0, 0, 0], [1.0, 1e-3, 0.0])
    assert st == po.ERR_INVALID_TWIST
    st, _ = po.model_f(po.MODEL_SIMPLE_CART, [0, 0, 0], [1.0, 1e-13, 0.0])
    assert st == po.OK


# ---- test/test_omni.cpp -----------------------------------------------------------------
def test_mecanum_kats(kats):
    k = kats["mecanum"]
    tol = k["tol"]
    st, f = po.model_f(po.MODEL_MECANUM, k["x"], k["u"], k["mp"])
    np.testing.assert_allclose(f, k["f"], atol=tol, rtol=0)
    st, A = po.model_fdx(po.MODEL_MECANUM, k["x"], k["u"], k["mp"])
    assert abs(A[0, 2] - k["fdx_02"]) < tol and abs(A[1, 2] - k["fdx_12"]) < tol
    st, B = po.model_fdu(po.MODEL_MECANUM, k["x"], k["mp"])
    np.testing.assert_allclose(B, np.array(k["fdu"]), atol=tol, rtol=0)


# ---- test/test_integrator.cpp -----------------------------------------------------------
def test_rk4_cart_straight_line(kats):
    k = kats["rk4_cart_straight"]
    T = po.steps(k["horizon"], k["dt"])
    assert T == 4
    ut = np.tile(np.array(k["u"])[:, None], (1, T))
    st, xt = po.rk4_solve_fwd(po.MODEL_CART, k["dt"], k["horizon"], k["x0"], ut, k["mp"])
    assert st == po.OK
    for i in range(T):
        assert ulp_close(xt[0, i], k["xt_x"][i]), (i, xt[0, i])
        assert ulp_close(xt[1, i], k["xt_y"][i])
        assert ulp_close(xt[2, i], k["xt_th"][i])


def test_integrate_twist_kats(kats):
    k = kats["integrate_twist_step"]
    pose = po.integrate_twist(k["x0"], k["vb"], k["dt"])
    np.testing.assert_allclose(pose, k["pose"], atol=k["tol"], rtol=0)
    k = kats["integrate_twist_traj"]
    x = np.array(k["x0"])
    for i in range(po.steps(k["horizon"], k["dt"])):
        x = po.integrate_twist(x, k["vb"], k["dt"])
        assert abs(x[0] - k["x"][i]) < k["tol"]
        assert abs(x[1] - k["y"][i]) < k["tol"]
        assert abs(x[2] - k["th"][i]) < k["tol"]


# ---- test/test_grid.cpp -----------------------------------------------------------------
def _grid(k, cells=None):
    b = k["bounds"]
    xs = po.lib().eo_axis_length(b[0], b[1], k["resolution"])
    ys = po.lib().eo_axis_length(b[2], b[3], k["resolution"])
    data = np.zeros(xs * ys, dtype=np.int8)
    for idx, v in (cells or {}).items():
        data[int(idx)] = v
    return po.GridMap(b[0], b[1], b[2], b[3], k["resolution"], data)


def test_grid_kats(kats):
    k = kats["grid_2x3"]
    g = _grid(k)
    assert (g.xsize, g.ysize) == (2, 3)
    assert g.grid2rowmajor(k["grid2rowmajor"]["i"], k["grid2rowmajor"]["j"]) == k["grid2rowmajor"]["idx"]
    assert g.rowmajor2grid(k["rowmajor2grid"]["idx"]) == (k["rowmajor2grid"]["i"], k["rowmajor2grid"]["j"])
    for idx, ok in k["bounds_idx"]:
        assert g.bounds_idx(idx) == ok
    for i, j, ok in k["bounds_ij"]:
        assert g.bounds_ij(i, j) == ok

    k = kats["grid_2x3_half"]
    g = _grid(k, k["cells"])
    t = k["grid2world_ij"]
    xy = g.grid2world(t["i"], t["j"])
    assert ulp_close(xy[0], t["xy"][0]) and ulp_close(xy[1], t["xy"][1])
    t = k["grid2world_idx"]
    xy = g.grid2world(*g.rowmajor2grid(t["idx"]))
    assert ulp_close(xy[0], t["xy"][0]) and ulp_close(xy[1], t["xy"][1])
    t = k["world2grid"]
    assert g.world2grid(t["x"], t["y"]) == (t["i"], t["j"])
    t = k["world2rowmajor"]
    assert g.grid2rowmajor(*g.world2grid(t["x"], t["y"])) == t["idx"]
    assert g.get_cell(k["get_cell_idx"]["idx"]) == k["get_cell_idx"]["val"]
    t = k["get_cell_xy"]
    assert g.get_cell(g.grid2rowmajor(*g.world2grid(t["x"], t["y"]))) == t["val"]
    t = k["get_cell_ij"]
    assert g.get_cell(g.grid2rowmajor(t["i"], t["j"])) == t["val"]
    with pytest.raises(ValueError):
        g.get_cell(6)
    with pytest.raises(ValueError):
        po.GridMap(0, 2, 0, 3, 1.0, np.zeros(5, dtype=np.int8))


# ---- SURVEY.md 8(c) anchors: outputs of the reference's sources ------------------------
ANCHOR_TOL = 1e-12  # stated oracle-vs-reference bar (Armadillo-internal orders unpinned below it)


def _closed_loop(anchors, key):
    c, a = anchors["closed_loop_common"], anchors[key]
    model = {"omni": po.MODEL_OMNI, "simple_cart": po.MODEL_SIMPLE_CART}[a["model"]]
    lim = np.array(a["limits"])
    ec = po.ErgodicControl(model, c["dt"], a["horizon"], c["target_resolution"], c["expl_weight"],
                           a["num_basis"], np.diag(a["Rinv_diag"]), -lim, lim)
    ec.set_target(c["means"], c["sigmas"])
    x = np.array(c["x0"])
    us = []
    for _ in range(len(a["u"])):
        u = ec.control(c["map_bounds"], x)
        us.append(u)
        st, x = po.rk4_step_fwd(model, c["dt"], x, u)
        assert st == po.OK
    return np.array(us), np.array(a["u"])


def test_anchor_omni_closed_loop(anchors):
    got, exp = _closed_loop(anchors, "omni_K10_T50")
    assert np.abs(got[0] - exp[0]).max() < ANCHOR_TOL
    # later calls feed back through the loop; rounding differences grow ~10x per call
    assert np.abs(got - exp).max() < 1e-11


def test_anchor_simple_cart_closed_loop(anchors):
    got, exp = _closed_loop(anchors, "simple_cart_K10_T20")
    assert np.abs(got - exp).max() < ANCHOR_TOL


def test_anchor_phik(anchors):
    a = anchors["phik_K10_121x61_trans0"]
    g = po.phi_grid(a["nx"], a["ny"], a["resolution"])
    assert g[0, a["nx"] - 1] == a["last_grid_x"]  # coordinates by accumulation, not j*res
    pv = po.target_fill(a["means"], a["sigmas"], a["trans"], g)
    assert abs(pv.sum() - 1.0) < 1e-13
    pk = po.spatial_coeff(a["lx"], a["ly"], a["num_basis"], pv, g)
    K = a["num_basis"]
    assert abs(pk[0] - a["phik_0"]) < 1e-13
    assert abs(pk[1] - a["phik_1"]) < 1e-13
    assert abs(pk[K] - a["phik_K"]) < 1e-13


def test_anchor_memory_path(anchors):
    a = anchors["memory_omni_K5"]
    lim = np.array(a["limits"])
    ec = po.ErgodicControl(po.MODEL_OMNI, a["dt"], a["horizon"], a["target_resolution"], 1.0,
                           a["num_basis"], np.diag(a["Rinv_diag"]), -lim, lim)
    ec.set_target(a["means"], a["sigmas"])
    u = ec.control(a["map_bounds"], a["x"], np.array(a["memory"]).T)
    assert np.abs(u - np.array(a["u"])).max() < ANCHOR_TOL


def test_anchor_scalars(anchors):
    s = anchors["scalars"]
    for rad, exp in s["normalize_angle_PI"]:
        assert abs(po.normalize_angle_PI(rad) - exp) < 1e-15
    for hor, dt, T in s["steps"]:
        assert po.steps(hor, dt) == T
    w = s["world2grid_wrap"]
    b = w["bounds"]
    xs = po.lib().eo_axis_length(b[0], b[1], w["resolution"])
    ys = po.lib().eo_axis_length(b[2], b[3], w["resolution"])
    g = po.GridMap(b[0], b[1], b[2], b[3], w["resolution"], np.zeros(xs * ys, dtype=np.int8))
    assert g.world2grid(w["x"], 0.0)[1] == w["j"]
    assert g.world2grid(w["x_big"], 0.0)[1] == w["j_big"]


def test_horizon_equal_dt_rejected():
    # ergodic_control.hpp:212-216
    with pytest.raises(ValueError):
        po.ErgodicControl(po.MODEL_OMNI, 0.1, 0.1, 0.1, 1.0, 5, np.eye(3), [-1] * 3, [1] * 3)


def test_missing_target_gives_nan():
    # skipping setTarget yields u = NaN (phi = 0/0)
    ec = po.ErgodicControl(po.MODEL_OMNI, 0.1, 1.0, 0.1, 1.0, 5, np.eye(3), [-1] * 3, [1] * 3)
    ec.set_target(np.zeros((0, 2)), np.zeros((0, 2)))
    u = ec.control((0, 12, 0, 6), [1, 1, 0.3])
    assert np.isnan(u).any()


# ---- per-stage fixtures and the independent numpy restatement (round 2) -------------------------------------
import glob as _glob

STAGE_FIXTURES = sorted(_glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stages_*.npz")))


def _load_stage_fixture(path):
    z = np.load(path)
    return {k: z[k] for k in z.files}


def test_stage_fixtures_exist_for_every_survey_config():
    names = {os.path.basename(p)[len("stages_"):-len(".npz")] for p in STAGE_FIXTURES}
    assert {"cfg1_omni_K5_T5", "yaml_omni_K10_T50", "yaml_cart_K10_T50", "cfg2_cart_K10_T20",
            "metric_cart_K10_T200", "metric_omni_K10_T200_mem", "cfg3_omni_K20_T250"} <= names


@pytest.mark.parametrize("path", STAGE_FIXTURES, ids=[os.path.basename(p)[7:-4] for p in STAGE_FIXTURES])
def test_numpy_restatement_reproduces_stage_fixtures(path):
    """The second restatement (tests/np_restatement.py, written from the reference's lines) against the
    fixtures the C oracle generated (tools/gen_golden.py): every stage of three consecutive control() calls to
    1e-12, and phi_k of the configuration (Target::fill + spatialCoeff) to 1e-12."""
    from tests import np_restatement as nr
    f = _load_stage_fixture(path)
    dt, horizon, res, w, K = f["params"]
    K = int(K)
    model = nr.Omni() if str(f["model"]) == "omni" else nr.SimpleCart()
    Rinv = np.diag(f["Rinv_diag"])
    lim = f["limits"]
    bounds = tuple(f["bounds"])
    phik, _, _, _ = nr.config_target_phik(bounds, res, K, f["means"], f["sigmas"])
    assert np.abs(phik - f["phik"]).max() <= 1e-12
    assert np.abs(nr.Basis(bounds[1] - bounds[0], bounds[3] - bounds[2], K).lamdak - f["lamdak"]).max() <= 1e-15
    mem = f["mem_cols"] if f["mem_cols"].shape[1] else None
    for call in range(3):
        st = nr.control_stages(model, dt, horizon, w, K, Rinv, -lim, lim, bounds, f["phik"], f["pose"],
                               f["ut_in_%d" % call], mem)
        for k in ("traj", "ck", "edx", "bdx", "rhot", "ut", "u0"):
            d = st[k] - f["%s_%d" % (k, call)]
            if k == "traj":
                d[2] = (d[2] + np.pi) % (2 * np.pi) - np.pi
            scale = max(1.0, float(np.abs(f["%s_%d" % (k, call)]).max()))
            assert np.abs(d).max() <= 1e-12 * scale, (k, call, float(np.abs(d).max()))
        # the chain of calls in the fixture is the oracle's own: ut after call n is ut_in of call n + 1
        if call < 2:
            assert np.array_equal(f["ut_%d" % call], f["ut_in_%d" % (call + 1)])


def test_stage_fixtures_are_what_the_oracle_produces_now():
    """tools/gen_golden.py is deterministic: regenerating one fixture in memory gives the committed bytes' values
    (guards against an oracle edit that silently changes results without regenerating / re-pinning)."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("gen_golden", os.path.join(root, "tools", "gen_golden.py"))
    gg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gg)
    import tempfile
    keep = gg.GOLDEN
    with tempfile.TemporaryDirectory() as tmp:
        gg.GOLDEN = tmp
        try:
            out = gg.generate("yaml_omni_K10_T50")
        finally:
            gg.GOLDEN = keep
    f = _load_stage_fixture(os.path.join(keep, "stages_yaml_omni_K10_T50.npz"))
    for k in ("phik", "traj_0", "ck_1", "rhot_2", "ut_2", "u0_2"):
        assert np.array_equal(out[k], f[k]), k
