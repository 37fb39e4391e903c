"""The HIP path (through the C ABI) against the COMMITTED per-stage fixtures tests/golden/stages_*.npz
(tools/gen_golden.py; cross-checked on CPU by the independent numpy restatement, tests/test_oracle_pinning.py):
phi_k of the configuration and every stage of three consecutive control() calls, each call started from the
fixture's own warm-start controls.  Tolerances: c_k, phi_k <= 1e-11; everything else <= 1e-9."""
import glob
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from ergodic_exploration_amd import capi

pytestmark = pytest.mark.gpu

FIXTURES = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stages_*.npz")))


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64).cuda()


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[7:-4] for p in FIXTURES])
def test_engine_reproduces_stage_fixtures(path):
    z = np.load(path)
    f = {k: z[k] for k in z.files}
    dt, horizon, res, w, K = f["params"]
    K = int(K)
    model = capi.MODEL_OMNI if str(f["model"]) == "omni" else capi.MODEL_SIMPLE_CART
    lim = f["limits"]
    eng = capi.Engine(capi.make_config(model, dt, horizon, res, w, K, np.diag(f["Rinv_diag"]), -lim, lim))
    eng.set_target_gaussians(f["means"], f["sigmas"])
    assert eng.config_domain(tuple(f["bounds"]))
    assert np.abs(eng.phik() - f["phik"]).max() <= 1e-11
    assert np.abs(eng.lamdak() - f["lamdak"]).max() <= 1e-15
    T, K2 = eng.T, eng.K2
    n_mem = f["mem_cols"].shape[1]
    d_mem = dev(f["mem_cols"].T.reshape(1, n_mem, 3)) if n_mem else None
    d_pose = dev(f["pose"].reshape(1, 3))
    d_u0 = torch.empty((1, 3), dtype=torch.float64, device="cuda")
    d_ck = torch.empty((1, K2), dtype=torch.float64, device="cuda")
    outs = {k: torch.empty((1, T, 3), dtype=torch.float64, device="cuda") for k in ("traj", "edx", "bdx", "rhot")}
    for call in range(3):
        d_ut = dev(f["ut_in_%d" % call].T.reshape(1, T, 3))
        eng.control_batch(1, d_pose, d_ut, d_u0, mem_cols=d_mem, mem_stride=n_mem, ck=d_ck, **outs)
        torch.cuda.synchronize()
        got = {k: v[0].cpu().numpy().T for k, v in outs.items()}
        got["ut"] = d_ut[0].cpu().numpy().T
        assert np.abs(d_ck[0].cpu().numpy() - f["ck_%d" % call]).max() <= 1e-11
        for k in ("traj", "edx", "bdx", "rhot", "ut"):
            d = got[k] - f["%s_%d" % (k, call)]
            if k == "traj":
                d[2] = (d[2] + np.pi) % (2 * np.pi) - np.pi
            assert np.abs(d).max() <= 1e-9, (k, call, float(np.abs(d).max()))
        assert np.abs(d_u0[0].cpu().numpy() - f["u0_%d" % call]).max() <= 1e-9
    eng.close()
