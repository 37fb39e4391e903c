"""EEA_OPT_RESIDENT_CONTROL (round 5): one robot, one control() per tick (reference exploration.hpp:232)
served by a RESIDENT workgroup through a host-mapped mailbox instead of one launch per call
(csrc/control_kernel_impl.hpp control_resident_kernel).  Same body as the launch path (control_agent), so the bar is bitwise
equality with it, plus the SURVEY.md 8(c) anchors through the resident path; and the life cycle: it leaves by itself after
250 ms without a call and is launched again, it restarts when phi_k / the domain changes, eea_set_ut / eea_get_ut /
eea_opt_traj see and are seen by it, the SimpleCart throw arrives as EEA_ERR_INVALID_TWIST, eea_destroy while it is there."""
import json
import os
import time

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from oracle import pyoracle as po
from ergodic_exploration_amd import capi
from tests.gpu_util import MAP_BOUNDS, make_pair

pytestmark = pytest.mark.gpu


@pytest.fixture
def resident():
    capi.set_option(capi.OPT_RESIDENT_CONTROL, 1)
    yield
    capi.set_option(capi.OPT_RESIDENT_CONTROL, 0)


@pytest.mark.parametrize("model,K,horizon,dt,precision", [
    ("omni", 5, 0.5, 0.1, capi.PREC_F64),           # BASELINE configs[0]
    ("simple_cart", 10, 2.0, 0.1, capi.PREC_F64),   # configs[1]
    ("simple_cart", 10, 20.0, 0.1, capi.PREC_F64),  # the metric shape (256 threads)
    ("omni", 20, 5.0, 0.02, capi.PREC_F32),         # configs[2]
    ("omni", 7, 3.0, 0.1, capi.PREC_F64),           # run-time K instance, 64 threads
    ("omni", 30, 50.0, 0.1, capi.PREC_F64),         # configs[4]'s control call: T = 500 (chunked scans), K = 30
])
def test_resident_path_is_bitwise_the_launch_path(model, K, horizon, dt, precision):
    """30 dependent calls with a growing replay memory (0 .. 29 columns) along a random walk: u0 of every call and the final
    warm start bitwise equal, launch path vs resident workgroup"""
    rng = np.random.default_rng(K + int(horizon))
    walk = np.array([3.0, 2.0, 0.3]) + np.cumsum(rng.normal(scale=0.03, size=(30, 3)), axis=0)
    out = {}
    for mode in (0, 1):
        capi.set_option(capi.OPT_RESIDENT_CONTROL, mode)
        try:
            eng, _ = make_pair(model, K, horizon, dt=dt, n_oracles=0, precision=precision)
            us = []
            for t in range(30):
                us.append(eng.control(MAP_BOUNDS, walk[t], walk[:t].T if t else None))
            out[mode] = (np.array(us), eng.get_ut(), eng.opt_traj())
            eng.close()
        finally:
            capi.set_option(capi.OPT_RESIDENT_CONTROL, 0)
    for a, b in zip(out[0], out[1]):
        assert np.array_equal(a, b)


def test_anchors_through_the_resident_path(resident):
    """closed-loop controls of the reference's own sources, through eea_control served by the resident
    workgroup (the same assertions as test_survey_anchors_through_c_abi)"""
    anchors = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "survey_anchors.json")))
    c = anchors["closed_loop_common"]
    for key in ("omni_K10_T50", "simple_cart_K10_T20"):
        a = anchors[key]
        cm = {"omni": capi.MODEL_OMNI, "simple_cart": capi.MODEL_SIMPLE_CART}[a["model"]]
        om = {"omni": po.MODEL_OMNI, "simple_cart": po.MODEL_SIMPLE_CART}[a["model"]]
        lim = np.array(a["limits"])
        eng = capi.Engine(capi.make_config(cm, c["dt"], a["horizon"], c["target_resolution"], c["expl_weight"], a["num_basis"],
                                           np.diag(a["Rinv_diag"]), -lim, lim))
        eng.set_target_gaussians(c["means"], c["sigmas"])
        x = np.array(c["x0"])
        for i, exp in enumerate(a["u"]):
            u = eng.control(c["map_bounds"], x)
            assert np.abs(u - np.array(exp)).max() < 1e-9 * (10 ** i), (key, i, u, exp)
            st, x = po.rk4_step_fwd(om, c["dt"], x, np.array(exp))
            assert st == po.OK
        eng.close()
    a = anchors["memory_omni_K5"]
    lim = np.array(a["limits"])
    eng = capi.Engine(capi.make_config(capi.MODEL_OMNI, a["dt"], a["horizon"], a["target_resolution"], 1.0, a["num_basis"],
                                       np.diag(a["Rinv_diag"]), -lim, lim))
    eng.set_target_gaussians(a["means"], a["sigmas"])
    u = eng.control(a["map_bounds"], a["x"], np.array(a["memory"]).T)
    assert np.abs(u - np.array(a["u"])).max() < 1e-9
    eng.close()


def test_resident_life_cycle(resident):
    eng, (orc,) = make_pair("omni", 10, 5.0)
    x = np.array([1.0, 1.0, 0.3])
    u = eng.control(MAP_BOUNDS, x)
    uo = orc.control(MAP_BOUNDS, x)
    assert np.abs(u - uo).max() < 1e-9
    # (a) it leaves after 250 ms without a call; the next call starts another one and is answered all the same
    time.sleep(0.6)
    for _ in range(3):
        u, uo = eng.control(MAP_BOUNDS, x), orc.control(MAP_BOUNDS, x)
        assert np.abs(u - uo).max() < 1e-8
    # (b) the warm start is visible to and from the other single-agent entry points
    assert np.abs(eng.get_ut() - orc.ut).max() < 1e-8
    tr, tro = eng.opt_traj(), orc.opt_traj()
    assert np.abs(tr[:2] - tro[:2]).max() < 1e-8
    ut = np.random.default_rng(0).uniform(-0.3, 0.3, (3, eng.T))
    eng.set_ut(ut)
    orc.ut = ut
    u, uo = eng.control(MAP_BOUNDS, x), orc.control(MAP_BOUNDS, x)
    assert np.abs(u - uo).max() < 1e-9
    # (c) the map grows: configTarget rebuilds phi_k (ergodic_control.hpp:362-416), the workgroup restarts with the new domain
    bigger = (-1.0, 13.0, -1.0, 5.0)
    u, uo = eng.control(bigger, x), orc.control(bigger, x)
    assert np.abs(u - uo).max() < 1e-8
    # ... and the map origin moves with an unchanged extent: map_pos_ is refreshed per call, nothing is rebuilt (:366-377)
    moved = (-0.5, 13.5, -1.5, 4.5)
    u, uo = eng.control(moved, x), orc.control(moved, x)
    assert np.abs(u - uo).max() < 1e-8
    # (d) a replay-memory sample beyond the mapped buffer takes the launch path for that call and comes back
    mem = np.tile(x[:, None], (1, 140)) + np.random.default_rng(1).normal(scale=0.2, size=(3, 140))
    u, uo = eng.control(moved, x, mem), orc.control(moved, x, mem)
    assert np.abs(u - uo).max() < 1e-8
    u, uo = eng.control(moved, x, mem[:, :50]), orc.control(moved, x, mem[:, :50])
    assert np.abs(u - uo).max() < 1e-8
    # (the reference's batch_size = 100 columns, and the full buffer of 128: two rounds of 64 columns in the one-wavefront server)
    for n in (100, 128, 1, 64, 65):
        u, uo = eng.control(moved, x, mem[:, :n]), orc.control(moved, x, mem[:, :n])
        assert np.abs(u - uo).max() < 1e-8, n
    eng.close()   # (e) eea_destroy with the workgroup resident
    # (f) SimpleCart::operator()'s throw (cart.hpp:167-170)
    eng, _ = make_pair("simple_cart", 10, 2.0, n_oracles=0)
    bad = np.zeros((3, eng.T))
    bad[1, 3] = 0.1
    eng.control(MAP_BOUNDS, x)
    eng.set_ut(bad)
    with pytest.raises(capi.EngineError) as ei:
        eng.control(MAP_BOUNDS, x)
    assert ei.value.status == capi.ERR_INVALID_TWIST
    assert np.array_equal(eng.get_ut(), bad)   # nothing of the agent was touched
    with pytest.raises(capi.EngineError):      # (the server is still there and still refuses: the reference keeps throwing)
        eng.control(MAP_BOUNDS, x)
    # ... and serves again once the controls are valid (the first request of a server was the refused one: its next one reads
    # the controls from memory, not from a hand-over that was never written)
    orc2 = make_pair("simple_cart", 10, 2.0, n_oracles=1)
    orc2[0].close()
    good = np.random.default_rng(3).uniform(-0.3, 0.3, (3, eng.T))
    good[1] = 0.0
    eng.set_ut(good)
    orc2[1][0].ut = good.copy()
    for _ in range(3):
        u, uo = eng.control(MAP_BOUNDS, x), orc2[1][0].control(MAP_BOUNDS, x)
        assert np.abs(u - uo).max() < 1e-8
    eng.close()


@pytest.mark.parametrize("model,K,horizon", [("omni", 10, 5.0),            # the one-wavefront server (T = 50)
                                             ("simple_cart", 10, 20.0)])   # the workgroup server (T = 200)
def test_calls_at_the_idle_period_are_served_once(model, K, horizon):
    """a request that arrives just as the server goes idle.  With the idle time cut to a few milliseconds
    (EEA_OPT_RESIDENT_IDLE_MS) and a call period swept ACROSS it in fine steps, many calls land in the window between the
    server's "alive = 0" and its last look at the mailbox.  Each must be served exactly once: the sequence of controls and the
    final warm start are bitwise those of the launch path (a request served twice shifts the warm start twice)."""
    idle_ms = 3
    rng = np.random.default_rng(11)
    n = 120
    walk = np.array([3.0, 2.0, 0.3]) + np.cumsum(rng.normal(scale=0.02, size=(n, 3)), axis=0)
    # periods from 0.6 to 1.4 idle times, and a stretch exactly around it in 10 us steps
    periods = np.concatenate([np.linspace(0.6, 1.4, 40) * idle_ms * 1e-3, idle_ms * 1e-3 + np.arange(-40, 40) * 1e-5])
    out = {}
    for mode in (0, 1):
        capi.set_option(capi.OPT_RESIDENT_CONTROL, mode)
        capi.set_option(capi.OPT_RESIDENT_IDLE_MS, idle_ms)
        try:
            eng, _ = make_pair(model, K, horizon, n_oracles=0)
            us = []
            for t in range(n):
                us.append(eng.control(MAP_BOUNDS, walk[t]))
                if mode:   # (the launch path needs no pacing)
                    t_end = time.perf_counter() + periods[t]
                    while time.perf_counter() < t_end:
                        pass
            out[mode] = (np.array(us), eng.get_ut())
            eng.close()
        finally:
            capi.set_option(capi.OPT_RESIDENT_CONTROL, 0)
            capi.set_option(capi.OPT_RESIDENT_IDLE_MS, 250)
    assert np.array_equal(out[0][0], out[1][0])
    assert np.array_equal(out[0][1], out[1][1])


def test_resident_stop_and_idle_option(resident):
    """eea_resident_stop makes the server leave at once (device-wide waits are cheap again), the next call
    starts another one with the warm start intact; EEA_OPT_RESIDENT_IDLE_MS is validated and honoured"""
    with pytest.raises(capi.EngineError):
        capi.set_option(capi.OPT_RESIDENT_IDLE_MS, 0)
    with pytest.raises(capi.EngineError):
        capi.set_option(capi.OPT_RESIDENT_IDLE_MS, 60001)
    assert capi.lib().eea_get_option(capi.OPT_RESIDENT_IDLE_MS) == 250
    eng, (orc,) = make_pair("omni", 10, 5.0)
    x = np.array([1.0, 1.0, 0.3])
    eng.resident_stop()   # nothing resident yet: EEA_OK
    for _ in range(3):
        u, uo = eng.control(MAP_BOUNDS, x), orc.control(MAP_BOUNDS, x)
        assert np.abs(u - uo).max() < 1e-9
    # resident: a device-wide wait takes until the idle time-out ...
    capi.set_option(capi.OPT_RESIDENT_IDLE_MS, 2000)
    try:
        eng.resident_stop()
        eng.control(MAP_BOUNDS, x), orc.control(MAP_BOUNDS, x)    # (a new server with the 2 s idle time)
        eng.resident_stop()                                        # ... unless it is told to leave first
        t0 = time.perf_counter()
        torch.cuda.synchronize()
        assert time.perf_counter() - t0 < 0.5
        for _ in range(3):   # the warm start survived the stop
            u, uo = eng.control(MAP_BOUNDS, x), orc.control(MAP_BOUNDS, x)
            assert np.abs(u - uo).max() < 1e-8
        eng.resident_stop()
        eng.resident_stop()   # idempotent
    finally:
        capi.set_option(capi.OPT_RESIDENT_IDLE_MS, 250)
    # a short idle time is honoured: after 3 x the idle time the server is gone and a device-wide wait returns at once
    capi.set_option(capi.OPT_RESIDENT_IDLE_MS, 20)
    try:
        eng.control(MAP_BOUNDS, x), orc.control(MAP_BOUNDS, x)
        time.sleep(0.06)
        t0 = time.perf_counter()
        torch.cuda.synchronize()
        assert time.perf_counter() - t0 < 0.015
        u, uo = eng.control(MAP_BOUNDS, x), orc.control(MAP_BOUNDS, x)
        assert np.abs(u - uo).max() < 1e-8
    finally:
        capi.set_option(capi.OPT_RESIDENT_IDLE_MS, 250)
    eng.close()
