"""GPU parity of the phi_k path (Target::fill + Basis::spatialCoeff behind configTarget) and of
the Basis free operations against the CPU oracle.  fp64 tolerance: <= 1e-11 abs on phi_k / c_k
(SURVEY.md 8(d)); the normalised target grid <= 1e-14 abs."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from oracle import pyoracle as po
from ergodic_exploration_amd import capi
from tests.gpu_util import MAP_BOUNDS, MEANS, SIGMAS

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["axis_factors", "fill_and_stream"])
def rebuild_impl(request):
    """configTarget of a Gaussian target has two implementations (EEA_OPT_REBUILD_IMPL): the per-axis factors in one launch
    (default) and Target::fill + the streaming Basis::spatialCoeff that explicit grids take -- both against the oracle"""
    capi.set_option(capi.OPT_REBUILD_IMPL, 1 if request.param == "fill_and_stream" else 0)
    yield request.param
    capi.set_option(capi.OPT_REBUILD_IMPL, 0)


def _engine(K, resolution=0.1, precision=capi.PREC_F64):
    return capi.Engine(capi.make_config(capi.MODEL_OMNI, 0.1, 2.0, resolution, 1.0, K, np.eye(3),
                                        [-1] * 3, [1] * 3, precision=precision))


def _oracle_phik(K, bounds, means, sigmas, resolution=0.1):
    o = po.ErgodicControl(po.MODEL_OMNI, 0.1, 2.0, resolution, 1.0, K, np.eye(3), [-1] * 3, [1] * 3)
    o.set_target(means, sigmas)
    assert o.config_target(bounds) == 1
    return o


@pytest.mark.parametrize("K,bounds,means,sigmas", [
    (10, MAP_BOUNDS, MEANS, SIGMAS),                                     # 121 x 61 points
    (5, (0.0, 12.0, 0.0, 6.0), [[2.5, 2.5]], [[1.5, 1.5]]),              # config 1
    (20, (0.0, 25.5, 0.0, 25.5), [[6.0, 6.0], [19.0, 12.0]], [[3.0, 3.0], [3.0, 3.0]]),  # 256 x 256
    (7, (-3.0, 30.3, 2.0, 9.7), [[1.0, 4.0], [20.0, 8.0], [12.0, 3.0]],
     [[2.0, 1.0], [4.0, 2.5], [0.7, 0.9]]),                              # ragged 334 x 78, 3 Gaussians
])
def test_phik_matches_oracle(K, bounds, means, sigmas, rebuild_impl):
    eng = _engine(K)
    eng.set_target_gaussians(means, sigmas)
    assert eng.config_domain(bounds) is True
    o = _oracle_phik(K, bounds, means, sigmas)
    assert np.abs(eng.phik() - o.phik).max() < 1e-11
    assert np.abs(eng.lamdak() - o.lamdak).max() < 1e-15
    # Target::fill output
    lx, ly = bounds[1] - bounds[0], bounds[3] - bounds[2]
    pv, nx, ny = eng.target_grid()
    g = po.phi_grid(nx, ny, 0.1)
    assert nx == po.lib().eo_axis_length(0.0, lx, 0.1) + 1 and ny == po.lib().eo_axis_length(0.0, ly, 0.1) + 1
    ref = po.target_fill(means, sigmas, [bounds[0], bounds[2]], g)
    assert np.abs(pv - ref).max() < 1e-14
    assert abs(pv.sum() - 1.0) < 1e-12
    eng.close()


@pytest.mark.parametrize("res,bounds,K", [
    (0.1, (0.0, 12.0, 0.0, 6.0), 10),       # 121 x 61: 1 grid point per thread of the fill kernel, one-tile direct finish
    (0.02, (0.0, 11.98, 0.0, 9.98), 6),     # 600 x 500 = 3.0e5 >= 2^18 points: 4 per thread
    (0.01, (-1.0, 20.0, 2.0, 23.0), 5),     # 2101 x 2101 = 4.4e6 >= 2^22 points: 16 per thread
])
def test_fill_kernel_regimes(res, bounds, K, rebuild_impl):
    """Target::fill takes 1 / 4 / 16 grid points per thread by grid size (and a grid of one tile is normalised by the
    streaming kernel itself): phi_k and the normalised grid against the oracle in each regime."""
    means, sigmas = [[2.5, 3.5], [8.5, 6.5]], [[1.5, 1.0], [0.8, 2.0]]
    eng = _engine(K, res)
    eng.set_target_gaussians(means, sigmas)
    assert eng.config_domain(bounds) is True
    o = _oracle_phik(K, bounds, means, sigmas, res)
    d = np.abs(eng.phik() - o.phik)
    # mode (0,0) is the mass ratio: the oracle adds the 4.4e6 values in one sequential chain (1e-11 of rounding there)
    assert d[1:].max() < 1e-11 and d[0] < 1e-10
    pv, nx, ny = eng.target_grid()
    lx, ly = bounds[1] - bounds[0], bounds[3] - bounds[2]
    assert nx == po.lib().eo_axis_length(0.0, lx, res) + 1 and ny == po.lib().eo_axis_length(0.0, ly, res) + 1
    ref = po.target_fill(means, sigmas, [bounds[0], bounds[2]], po.phi_grid(nx, ny, res))
    assert np.abs(pv - ref).max() < 1e-14
    eng.close()


def test_phik_anchor_from_reference(anchors, rebuild_impl):
    a = anchors["phik_K10_121x61_trans0"]
    eng = _engine(a["num_basis"], a["resolution"])
    eng.set_target_gaussians(a["means"], a["sigmas"])
    eng.config_domain((0.0, a["lx"], 0.0, a["ly"]))  # origin (0,0): trans = 0
    pk = eng.phik()
    K = a["num_basis"]
    assert abs(pk[0] - a["phik_0"]) < 1e-11
    assert abs(pk[1] - a["phik_1"]) < 1e-11
    assert abs(pk[K] - a["phik_K"]) < 1e-11
    eng.close()


def test_rebuild_rule():
    """configTarget rebuilds only when the extent changes by >= 1e-12; map_pos follows every
    call; a later setTarget alone does not rebuild (ergodic_control.hpp:366-377)."""
    eng = _engine(10)
    eng.set_target_gaussians(MEANS, SIGMAS)
    assert eng.config_domain(MAP_BOUNDS) is True
    pk = eng.phik()
    assert eng.config_domain(MAP_BOUNDS) is False
    assert eng.config_domain((0.0, 12.0, 0.0, 6.0)) is False      # origin moved, same extent
    eng.set_target_gaussians([[1.0, 1.0]], [[0.5, 0.5]])
    assert eng.config_domain((0.0, 12.0, 0.0, 6.0)) is False
    assert np.array_equal(eng.phik(), pk)
    assert eng.config_domain((0.0, 12.0 + 5e-13, 0.0, 6.0)) is False
    assert eng.config_domain((0.0, 12.5, 0.0, 6.0)) is True       # map grew
    o = _oracle_phik(10, (0.0, 12.5, 0.0, 6.0), [[1.0, 1.0]], [[0.5, 0.5]])
    assert np.abs(eng.phik() - o.phik).max() < 1e-11
    eng.close()


@pytest.mark.parametrize("K,bounds", [(10, (-1.0, 11.0, -1.0, 5.0)), (20, (0.0, 25.5, 0.0, 25.5))])
def test_rebuild_enqueued_only(K, bounds, rebuild_impl):
    """eea_config_domain_async: the rebuild is only enqueued.  Control calls on the SAME stream are ordered by the
    stream, control calls on ANOTHER stream are made to wait by the engine, the getters wait: phi_k and the controls
    are bitwise those of the synchronous form."""
    means, sigmas = [[2.5, 2.5], [8.5, 2.5]], [[1.5, 1.5], [1.5, 1.5]]
    rng = np.random.default_rng(4)
    B = 64
    res = {}
    for form in ("sync", "async_same_stream", "async_other_stream"):
        eng = capi.Engine(capi.make_config(capi.MODEL_OMNI, 0.1, 2.0, 0.1, 1.0, K, np.diag([1.0, 1.0, 2.0]), [-1, -1, -2], [1, 1, 2]))
        eng.set_target_gaussians(means, sigmas)
        T = eng.T
        rs = np.random.default_rng(9)
        poses = np.stack([rs.uniform(bounds[0] + 1, bounds[1] - 1, B), rs.uniform(bounds[2] + 1, bounds[3] - 1, B),
                          rs.uniform(-3, 3, B)], 1)
        d_pose = torch.as_tensor(poses).cuda()
        d_ut = torch.zeros((B, T, 3), dtype=torch.float64, device="cuda")
        d_u0 = torch.empty((B, 3), dtype=torch.float64, device="cuda")
        sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
        torch.cuda.synchronize()
        for step in range(3):   # the map grows on every step: a rebuild each time, then a control pass
            b = (bounds[0], bounds[1] + 0.1 * step, bounds[2], bounds[3])
            if form == "sync":
                assert eng.config_domain(b, stream=sa.cuda_stream) is True
                eng.control_batch(B, d_pose, d_ut, d_u0, stream=sa.cuda_stream)
            elif form == "async_same_stream":
                assert eng.config_domain_async(b, stream=sa.cuda_stream) is True
                eng.control_batch(B, d_pose, d_ut, d_u0, stream=sa.cuda_stream)
            else:
                sa.wait_stream(sb)     # the caller's part: the rebuild follows the control pass still reading phi_k
                assert eng.config_domain_async(b, stream=sa.cuda_stream) is True
                eng.control_batch(B, d_pose, d_ut, d_u0, stream=sb.cuda_stream)
        pk = eng.phik()                # waits for the last rebuild
        torch.cuda.synchronize()
        res[form] = (pk, d_u0.cpu().numpy().copy(), d_ut.cpu().numpy().copy())
        eng.close()
    for form in ("async_same_stream", "async_other_stream"):
        for a, b_ in zip(res["sync"], res[form]):
            assert np.array_equal(a, b_), form
    assert np.isfinite(res["sync"][1]).all()


def test_set_target_grid_entropy_surrogate():
    """BASELINE config 5 entry (reduced): explicit target grid from an int8 occupancy map through
    entropy() (numerics.hpp:164-179), normalised to sum 1 -> Basis::spatialCoeff."""
    rng = np.random.default_rng(2024)
    nx = ny = 129
    res = 0.1
    lx = ly = (nx - 1) * res
    blocks = rng.choice(np.array([0, 100, -1], dtype=np.int8), size=(ny // 16 + 1, nx // 16 + 1), p=[0.7, 0.1, 0.2])
    occ = np.kron(blocks, np.ones((16, 16), dtype=np.int8))[:ny, :nx]
    ent = np.array([po.lib().eo_entropy(float(v) / 100.0) for v in occ.reshape(-1)])
    phi = ent / ent.sum()
    K = 12
    eng = _engine(K, res)
    eng.set_target_grid(nx, ny, torch.as_tensor(phi).cuda(), lx, ly)
    g = po.phi_grid(nx, ny, res)
    ref = po.spatial_coeff(lx, ly, K, phi, g)
    assert np.abs(eng.phik() - ref).max() < 1e-11
    # host-pointer form, and linearity of spatialCoeff in phi_vals
    eng.set_target_grid(nx, ny, 2.0 * phi, lx, ly)
    assert np.abs(eng.phik() - 2.0 * ref).max() < 2e-11
    eng.close()


def test_phik_f32():
    eng = _engine(20, precision=capi.PREC_F32)
    bounds = (0.0, 25.5, 0.0, 25.5)
    means, sigmas = [[6.0, 6.0], [19.0, 12.0]], [[3.0, 3.0], [3.0, 3.0]]
    eng.set_target_gaussians(means, sigmas)
    eng.config_domain(bounds)
    o = _oracle_phik(20, bounds, means, sigmas)
    assert np.abs(eng.phik() - o.phik).max() < 2e-5
    eng.close()


def test_large_grid_property():
    """1024 x 1024 grid, K = 30 (config 5 size): phi_k(0,0) = sum(phi) = 1, and the separable
    kernel agrees with the point-list kernel (different code path) to 1e-11."""
    nx = ny = 1024
    res = 0.1
    lx = ly = (nx - 1) * res
    rng = np.random.default_rng(7)
    phi = rng.random(nx * ny)
    phi /= phi.sum()
    K = 30
    eng = _engine(K, res)
    eng.set_target_grid(nx, ny, torch.as_tensor(phi).cuda(), lx, ly)
    pk = eng.phik()
    assert abs(pk[0] - 1.0) < 1e-12
    g = po.phi_grid(nx, ny, res)
    pk2 = capi.basis_spatial_coeff(lx, ly, K, phi, g)
    assert np.abs(pk - pk2).max() < 1e-11
    eng.close()


def test_basis_free_functions():
    rng = np.random.default_rng(3)
    lx, ly, K = 12.0, 6.0, 10
    xt = np.vstack([rng.uniform(0, lx, 333), rng.uniform(0, ly, 333), rng.uniform(-3, 3, 333)])
    assert np.abs(capi.basis_traj_coeff(lx, ly, K, xt) - po.traj_coeff(lx, ly, K, xt)).max() < 1e-12
    P = 5000
    grid = np.vstack([rng.uniform(0, lx, P), rng.uniform(0, ly, P)])
    pv = rng.random(P)
    assert np.abs(capi.basis_spatial_coeff(lx, ly, K, pv, grid) -
                  po.spatial_coeff(lx, ly, K, pv, grid)).max() < 1e-10
    # single point == fourierBasis
    x = np.array([[3.3], [1.7]])
    assert np.abs(capi.basis_traj_coeff(lx, ly, 5, x) - po.fourier_basis(lx, ly, 5, x[:, 0])).max() < 1e-15


def test_row_tiles_sum_to_full_grid():
    """Grid-tiled phi_k (config 5 sharding): partials of row tiles add up to the full-grid result,
    and installing the sum with eea_set_phik drives control like the single-GPU path."""
    from ergodic_exploration_amd import agent_batch as ab
    nx, ny, K, res = 300, 257, 12, 0.1
    lx, ly = (nx - 1) * res, (ny - 1) * res
    rng = np.random.default_rng(11)
    phi = rng.random(nx * ny)
    phi /= phi.sum()
    d_phi = torch.as_tensor(phi).cuda()
    eng = _engine(K, res)
    eng.set_target_grid(nx, ny, d_phi, lx, ly)
    full = eng.phik()
    total = torch.zeros(K * K, dtype=torch.float64, device="cuda")
    for world in (2, 3, 8):
        total.zero_()
        for rank in range(world):
            row0, nrows = ab.grid_row_tile(ny, rank, world)
            part = torch.empty(K * K, dtype=torch.float64, device="cuda")
            eng.spatial_coeff_rows(nx, ny, row0, nrows, d_phi[row0 * nx:(row0 + nrows) * nx], lx, ly, part)
            total += part
        torch.cuda.synchronize()
        assert np.abs(total.cpu().numpy() - full).max() < 1e-13
    eng2 = _engine(K, res)
    eng2.set_phik(total, lx, ly)
    assert np.array_equal(eng2.phik(), total.cpu().numpy())
    u1 = eng.control((0.0, lx, 0.0, ly), [3.0, 4.0, 0.5])
    u2 = eng2.control((0.0, lx, 0.0, ly), [3.0, 4.0, 0.5])
    assert np.abs(u1 - u2).max() < 1e-9
    eng.close()
    eng2.close()


def _occupancy(nx, ny, seed=2024, block=32):
    """BASELINE config 5 occupancy: 70 % free (0), 10 % occupied (100), 20 % unknown (-1) in blocks"""
    rng = np.random.default_rng(seed)
    blocks = rng.choice(np.array([0, 100, -1], dtype=np.int8), size=(ny // block + 1, nx // block + 1),
                        p=[0.7, 0.1, 0.2])
    return np.ascontiguousarray(np.kron(blocks, np.ones((block, block), dtype=np.int8))[:ny, :nx])


def _entropy_target(occ):
    lut = np.array([po.lib().eo_entropy(float(np.int8(np.uint8(b))) / 100.0) for b in range(256)])
    ent = lut[occ.reshape(-1).view(np.uint8)]
    return ent / ent.sum()


@pytest.mark.parametrize("nx,ny,K", [(129, 129, 12), (300, 77, 10), (257, 513, 30), (64, 40, 5)])
def test_occupancy_target_fused(nx, ny, K):
    """eea_set_target_occupancy (int8 cells -> entropy -> normalise -> spatialCoeff in one streaming
    pass) against the oracle's spatialCoeff on the materialised entropy target."""
    res = 0.1
    lx, ly = (nx - 1) * res, (ny - 1) * res
    occ = _occupancy(nx, ny, block=16)
    # a few raw probabilities as well (cells strictly between 0 and 100)
    occ[3, 5:9] = [17, 50, 83, 99]
    ref = po.spatial_coeff(lx, ly, K, _entropy_target(occ), po.phi_grid(nx, ny, res))
    eng = _engine(K, res)
    eng.set_target_occupancy(nx, ny, torch.as_tensor(occ).cuda(), lx, ly)
    assert np.abs(eng.phik() - ref).max() < 1e-11
    assert abs(eng.phik()[0] - 1.0) < 1e-14  # normalised: mode (0,0) is the total mass
    eng.set_target_occupancy(nx, ny, occ, lx, ly)  # host-pointer form
    assert np.abs(eng.phik() - ref).max() < 1e-11
    with pytest.raises(capi.EngineError):  # the fp64 target grid is never materialised
        eng.target_grid()
    eng.close()


def test_occupancy_row_tiles():
    """config 5 sharding: un-normalised sums of row tiles add up; total / total[0] == fused result"""
    from ergodic_exploration_amd import agent_batch as ab
    nx, ny, K, res = 200, 131, 10, 0.1
    lx, ly = (nx - 1) * res, (ny - 1) * res
    occ = _occupancy(nx, ny, block=8)
    d_occ = torch.as_tensor(occ).cuda()
    eng = _engine(K, res)
    eng.set_target_occupancy(nx, ny, d_occ, lx, ly)
    full = eng.phik()
    for world in (2, 3, 8):
        total = torch.zeros(K * K, dtype=torch.float64, device="cuda")
        for rank in range(world):
            row0, nrows = ab.grid_row_tile(ny, rank, world)
            part = torch.empty(K * K, dtype=torch.float64, device="cuda")
            eng.spatial_coeff_occupancy_rows(nx, ny, row0, nrows, d_occ[row0:row0 + nrows], lx, ly, part)
            total += part
        torch.cuda.synchronize()
        assert np.abs((total / total[0]).cpu().numpy() - full).max() < 1e-13
        # the device-side finish of the tiled form: phi_k = sums / sums[0], stream-ordered, no host round trip
        e2 = _engine(K, res)
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            sums = total.clone()
        e2.set_phik_from_sums(sums, lx, ly, stream=st.cuda_stream)
        torch.cuda.synchronize()
        assert np.abs(e2.phik() - full).max() < 1e-13
        e2.close()
    eng.close()


def test_occupancy_target_f32_and_config5_size():
    """fp32 engine on the full BASELINE config 5 grid (1024 x 1024, K = 30) against the fp64 engine"""
    nx = ny = 1024
    res = 0.1
    lx = ly = 102.4
    occ = torch.as_tensor(_occupancy(nx, ny)).cuda()
    e64 = _engine(30, res)
    e64.set_target_occupancy(nx, ny, occ, lx, ly)
    e32 = _engine(30, res, precision=capi.PREC_F32)
    e32.set_target_occupancy(nx, ny, occ, lx, ly)
    assert np.abs(e32.phik() - e64.phik()).max() < 2e-5
    # the fp64 result against numpy on the same entropy target (separable form, fp64)
    phi = _entropy_target(occ.cpu().numpy()).reshape(ny, nx)
    xs = np.concatenate([[0.0], np.cumsum(np.full(nx - 1, res))])  # coordinates by accumulation
    ys = np.concatenate([[0.0], np.cumsum(np.full(ny - 1, res))])
    cx = np.cos(np.outer(np.arange(30) * (np.pi / lx), xs))
    cy = np.cos(np.outer(np.arange(30) * (np.pi / ly), ys))
    ref = (cy @ phi @ cx.T).reshape(-1)  # [k2][k1] -> col = k2*K + k1
    assert np.abs(e64.phik() - ref).max() < 1e-11
    e64.close()
    e32.close()


@pytest.mark.parametrize("seed", range(10))
def test_random_grid_shapes(seed):
    """Randomised grid shapes through both inputs of the streaming kernel (fp64 values, int8 cells):
    widths that break the 16-byte / 8-byte vector loads, grids narrower than one 16-column operand
    and shorter than one 4-row step, heights that leave a partial pipeline stage."""
    rng = np.random.default_rng(500 + seed)
    nx = int(rng.choice([1, 2, 3, 5, 15, 16, 17, 63, 66, 127, 130, 257]))
    ny = int(rng.choice([1, 2, 3, 4, 5, 31, 33, 64, 70, 129]))
    K = int(rng.choice([1, 2, 5, 7, 10, 16, 17, 20, 30, 32]))
    res = 0.1
    lx, ly = max(nx - 1, 1) * res, max(ny - 1, 1) * res
    g = po.phi_grid(nx, ny, res)
    phi = rng.random(nx * ny)
    ref = po.spatial_coeff(lx, ly, K, phi, g)
    eng = _engine(K, res)
    eng.set_target_grid(nx, ny, torch.as_tensor(phi).cuda(), lx, ly)
    assert np.abs(eng.phik() - ref).max() < 1e-11 * max(1.0, phi.sum())
    occ = rng.choice(np.array([0, 100, -1, 25, 77], dtype=np.int8), size=(ny, nx))
    tgt = _entropy_target(occ)
    ref_o = po.spatial_coeff(lx, ly, K, tgt, g)
    eng.set_target_occupancy(nx, ny, torch.as_tensor(occ).cuda(), lx, ly)
    assert np.abs(eng.phik() - ref_o).max() < 1e-11
    eng.close()
