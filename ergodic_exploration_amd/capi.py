"""ctypes view of the C ABI (include/ergodic_amd.h) of the gfx950 engine.

This is plumbing for tests and bench.py: it loads ergodic_exploration_amd/lib/libergodic_amd.so
and fails loudly when the library is missing.  There is no CPU fallback: every compute
entry point runs HIP kernels on the device.
"""
import ctypes as C
import os
import re
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
# EEA_LIB_VARIANT selects an A/B build of the same sources (csrc/Makefile VARIANT=...)
LIB_PATH = os.path.join(_HERE, "lib", "libergodic_amd%s.so" % os.environ.get("EEA_LIB_VARIANT", ""))
HEADER_PATH = os.path.join(ROOT, "include", "ergodic_amd.h")

MODEL_OMNI, MODEL_SIMPLE_CART = 0, 1
PREC_F64, PREC_F32 = 0, 1
OK, ERR_INVALID_ARGUMENT, ERR_INVALID_TWIST, ERR_UNSUPPORTED, ERR_HIP, ERR_NO_TARGET, ERR_TIMEOUT = range(7)
# eea_set_option (process-wide dispatch options; the library reads no environment variable)
OPT_CONTROL_KERNEL, OPT_WORKGROUP_THREADS, OPT_COLLISION_IMPL, OPT_MAILBOX_POLL, OPT_REBUILD_IMPL, OPT_AGENT_LANES, OPT_RESIDENT_CONTROL, OPT_RESIDENT_IDLE_MS = range(8)


class EngineError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("eea status %d: %s" % (status, msg))
        self.status = status


class Config(C.Structure):
    _fields_ = [("model", C.c_int), ("precision", C.c_int), ("device", C.c_int),
                ("dt", C.c_double), ("horizon", C.c_double), ("resolution", C.c_double),
                ("expl_weight", C.c_double), ("num_basis", C.c_uint),
                ("Rinv", C.c_double * 9), ("umin", C.c_double * 3), ("umax", C.c_double * 3)]


class BatchIO(C.Structure):
    _fields_ = [("d_pose", C.c_void_p), ("d_ut", C.c_void_p), ("d_mem_cols", C.c_void_p),
                ("d_n_mem", C.c_void_p), ("mem_stride", C.c_uint), ("d_u0", C.c_void_p),
                ("d_traj", C.c_void_p), ("d_ck", C.c_void_p), ("d_edx", C.c_void_p),
                ("d_bdx", C.c_void_p), ("d_rhot", C.c_void_p), ("d_status", C.c_void_p),
                ("d_ck_shared", C.c_void_p), ("d_ck_rec", C.c_void_p), ("ck_shared_parts", C.c_uint),
                ("d_rec_ready", C.c_void_p), ("rec_seq", C.c_uint), ("d_ck_flag", C.c_void_p), ("ck_flag_seq", C.c_uint),
                ("d_skip", C.c_void_p), ("rec_per_wavefront", C.c_int)]


class ConsensusDesc(C.Structure):
    _fields_ = [("n_groups", C.c_uint), ("group_agents", C.POINTER(C.c_uint)), ("group_io", C.POINTER(BatchIO)),
                ("lag", C.c_uint), ("passes_per_launch", C.c_uint)]


class TickIO(C.Structure):
    _fields_ = [("d_follow_dwa", C.c_void_p), ("d_dwa_count", C.c_void_p), ("d_u", C.c_void_p), ("d_vb", C.c_void_p),
                ("d_grid", C.c_void_p), ("d_traj", C.c_void_p), ("d_valid", C.c_void_p), ("d_skip", C.c_void_p),
                ("d_source", C.c_void_p), ("val_dt", C.c_double), ("val_horizon", C.c_double), ("grid_epoch", C.c_ulonglong)]


class CollisionCfg(C.Structure):
    _fields_ = [("xmin", C.c_double), ("ymin", C.c_double), ("resolution", C.c_double),
                ("xsize", C.c_uint), ("ysize", C.c_uint), ("boundary_radius", C.c_double),
                ("search_radius", C.c_double), ("obstacle_threshold", C.c_double),
                ("occupied_threshold", C.c_double)]


class DwaCfg(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("dt", "horizon", "acc_dt", "acc_lim_x", "acc_lim_y", "acc_lim_th",
                                          "max_vel_x", "min_vel_x", "max_vel_y", "min_vel_y",
                                          "max_rot_vel", "min_rot_vel")] + \
               [(n, C.c_uint) for n in ("vx_samples", "vy_samples", "vth_samples")]


def build(force=False):
    """hipcc --offload-arch=gfx950 build of the library (csrc/Makefile)."""
    args = ["make", "-s", "-j4", "-C", os.path.join(_HERE, "csrc")]
    if force:
        subprocess.check_call(args + ["clean"])
    subprocess.check_call(args)
    return LIB_PATH


def declared_symbols():
    """Entry points declared in include/ergodic_amd.h."""
    with open(HEADER_PATH) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(eea_[a-z0-9_]+)\s*\(", text)))


_lib = None


def lib():
    """Loads the shared library; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libergodic_amd.so is missing: run __graft_entry__.build() "
                               "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        # One HIP runtime per process: the tests and bench.py hand torch-allocated device
        # memory and streams to the engine, so torch's bundled libamdhip64 must be the
        # instance this library binds to (loading the system one first gives two runtimes).
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        L.eea_last_error.restype = C.c_char_p
        L.eea_steps.restype = C.c_uint
        L.eea_batch_agent_lanes.restype = C.c_uint
        L.eea_batch_agent_lanes.argtypes = [C.c_void_p, C.c_uint]
        L.eea_batch_record_count.restype = C.c_uint
        L.eea_batch_record_count.argtypes = [C.c_void_p, C.c_uint]
        L.eea_num_modes.restype = C.c_uint
        L.eea_real_size.restype = C.c_size_t
        L.eea_time_step.restype = C.c_double
        L.eea_abi_version.restype = C.c_uint
        L.eea_ck_record_len.restype = C.c_uint
        L.eea_ck_record_len.argtypes = [C.c_void_p]
        L.eea_ck_records_sum.argtypes = [C.c_void_p, C.c_uint, C.c_void_p, C.c_void_p, C.c_void_p]
        L.eea_ck_records_sum_bound.argtypes = [C.c_void_p, C.c_uint, C.c_void_p, C.c_void_p, C.c_uint, C.c_void_p, C.c_void_p,
                                               C.c_void_p]
        L.eea_publish_record.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint, C.c_void_p]
        L.eea_stream_wait_flag.argtypes = [C.c_void_p, C.c_uint, C.c_void_p, C.c_void_p]
        L.eea_consensus_plan_create.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
        L.eea_consensus_plan_launch.argtypes = [C.c_void_p, C.c_void_p]
        L.eea_consensus_plan_info.argtypes = [C.c_void_p, C.POINTER(C.c_uint), C.POINTER(C.c_void_p)]
        L.eea_consensus_plan_destroy.argtypes = [C.c_void_p]
        L.eea_consensus_plan_destroy.restype = None
        L.eea_ck_records_sum_ws_bytes.argtypes = [C.c_void_p, C.c_uint, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        L.eea_ck_records_sum_ws.argtypes = [C.c_void_p, C.c_uint, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        for name in ("eea_steps", "eea_num_modes", "eea_real_size", "eea_time_step", "eea_destroy", "eea_resident_stop"):
            getattr(L, name).argtypes = [C.c_void_p]
        L.eea_destroy.restype = None
        L.eea_create.argtypes = [C.POINTER(Config), C.POINTER(C.c_void_p)]
        L.eea_set_target_gaussians.argtypes = [C.c_void_p, C.c_uint, C.c_void_p, C.c_void_p]
        L.eea_set_target_grid.argtypes = [C.c_void_p, C.c_uint, C.c_uint, C.c_void_p, C.c_int,
                                          C.c_double, C.c_double, C.c_void_p]
        L.eea_spatial_coeff_rows.argtypes = [C.c_void_p, C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_void_p,
                                             C.c_double, C.c_double, C.c_void_p, C.c_void_p]
        L.eea_set_target_occupancy.argtypes = [C.c_void_p, C.c_uint, C.c_uint, C.c_void_p, C.c_int,
                                               C.c_double, C.c_double, C.c_void_p]
        L.eea_spatial_coeff_occupancy_rows.argtypes = [C.c_void_p, C.c_uint, C.c_uint, C.c_uint, C.c_uint,
                                                       C.c_void_p, C.c_double, C.c_double, C.c_void_p,
                                                       C.c_void_p]
        L.eea_set_phik.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_double]
        L.eea_set_phik_from_sums.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_void_p]
        L.eea_config_domain.argtypes = [C.c_void_p] + [C.c_double] * 4 + [C.POINTER(C.c_int), C.c_void_p]
        L.eea_config_domain_async.argtypes = [C.c_void_p] + [C.c_double] * 4 + [C.POINTER(C.c_int), C.c_void_p]
        L.eea_get_phik.argtypes = [C.c_void_p, C.c_void_p]
        L.eea_get_lamdak.argtypes = [C.c_void_p, C.c_void_p]
        L.eea_target_grid_size.argtypes = [C.c_void_p, C.POINTER(C.c_uint), C.POINTER(C.c_uint)]
        L.eea_get_target_grid.argtypes = [C.c_void_p, C.c_void_p]
        L.eea_control_batch.argtypes = [C.c_void_p, C.c_uint, C.POINTER(BatchIO), C.c_void_p]
        L.eea_control_batch_steps.argtypes = [C.c_void_p, C.c_uint, C.POINTER(BatchIO), C.c_uint, C.c_uint, C.c_uint,
                                              C.c_void_p]
        L.eea_set_option.argtypes = [C.c_int, C.c_int]
        L.eea_get_option.argtypes = [C.c_int]
        if hasattr(L, "eea_debug_phase_timing"):  # A/B library only (EEA_LIB_VARIANT=_ab, tools/ab/)
            L.eea_debug_phase_timing.argtypes = [C.c_void_p, C.c_uint, C.POINTER(BatchIO), C.c_void_p, C.c_void_p]
        L.eea_comm_get_unique_id.argtypes = [C.c_void_p]
        L.eea_comm_set_library.argtypes = [C.c_char_p]
        L.eea_comm_set_library.restype = C.c_int
        L.eea_comm_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]
        L.eea_comm_destroy.argtypes = [C.c_void_p]
        L.eea_comm_destroy.restype = None
        L.eea_comm_rank.argtypes = [C.c_void_p]
        L.eea_comm_nranks.argtypes = [C.c_void_p]
        L.eea_ck_sum.argtypes = [C.c_void_p, C.c_uint, C.c_void_p, C.c_void_p, C.c_void_p]
        L.eea_comm_allgather_ck.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_void_p, C.c_void_p, C.c_void_p]
        L.eea_comm_consensus_ck.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_void_p, C.c_void_p, C.c_void_p]
        L.eea_comm_allreduce_sum_async.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint, C.c_void_p, C.c_int]
        L.eea_comm_records_exchange_async.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_void_p, C.c_void_p,
                                                      C.c_void_p, C.c_uint, C.c_int]
        L.eea_comm_wait.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.eea_comm_records_exchange_bound.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_void_p, C.c_void_p, C.c_uint,
                                                      C.c_void_p, C.c_void_p, C.c_int]
        L.eea_comm_allreduce_sum.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint, C.c_void_p]
        L.eea_comm_consensus_ck_async.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.eea_comm_allgather_ck_async.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.eea_comm_wait.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.eea_rollout_batch.argtypes = [C.c_void_p, C.c_uint, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p]
        L.eea_control.argtypes = [C.c_void_p] + [C.c_double] * 4 + [C.c_void_p, C.c_void_p, C.c_uint,
                                                                     C.c_void_p]
        L.eea_opt_traj.argtypes = [C.c_void_p, C.c_void_p]
        L.eea_get_ut.argtypes = [C.c_void_p, C.c_void_p]
        L.eea_set_ut.argtypes = [C.c_void_p, C.c_void_p]
        L.eea_basis_traj_coeff.argtypes = [C.c_int, C.c_double, C.c_double, C.c_uint, C.c_void_p,
                                           C.c_uint, C.c_uint, C.c_void_p]
        L.eea_basis_spatial_coeff.argtypes = [C.c_int, C.c_double, C.c_double, C.c_uint, C.c_void_p,
                                              C.c_void_p, C.c_uint, C.c_void_p]
        L.eea_rk4_rollout.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
        L.eea_target_fill.argtypes = [C.c_int, C.c_uint, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint, C.c_void_p]
        L.eea_dwa_control_batch.argtypes = [C.c_int, C.POINTER(CollisionCfg), C.POINTER(DwaCfg), C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint,
                                            C.c_double, C.c_uint, C.c_void_p, C.c_void_p, C.c_void_p]
        L.eea_collision_check_batch.argtypes = [C.c_int, C.POINTER(CollisionCfg), C.c_void_p,
                                                C.c_void_p, C.c_uint, C.c_void_p, C.c_void_p]
        L.eea_validate_control_batch.argtypes = [C.c_int, C.POINTER(CollisionCfg), C.c_void_p,
                                                 C.c_void_p, C.c_void_p, C.c_double, C.c_double,
                                                 C.c_uint, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def check(status):
    if status != OK:
        raise EngineError(status, lib().eea_last_error().decode())


def make_config(model, dt, horizon, resolution, expl_weight, num_basis, Rinv, umin, umax,
                precision=PREC_F64, device=0):
    cfg = Config()
    cfg.model, cfg.precision, cfg.device = model, precision, device
    cfg.dt, cfg.horizon, cfg.resolution, cfg.expl_weight = dt, horizon, resolution, expl_weight
    cfg.num_basis = num_basis
    for c in range(3):
        for r in range(3):
            cfg.Rinv[r + 3 * c] = float(Rinv[r][c])
    for i in range(3):
        cfg.umin[i] = float(umin[i])
        cfg.umax[i] = float(umax[i])
    return cfg


def _ptr(t):
    """Raw device/host pointer of a torch tensor, numpy array or None."""
    if t is None:
        return None
    if hasattr(t, "data_ptr"):
        return C.c_void_p(t.data_ptr())
    return C.c_void_p(t.ctypes.data)


class Engine:
    """One `ErgodicControl` engine (handle of the C ABI)."""

    def __init__(self, cfg):
        self.cfg = cfg
        self.h = C.c_void_p()
        check(lib().eea_create(C.byref(cfg), C.byref(self.h)))
        self.T = int(lib().eea_steps(self.h))
        self.K2 = int(lib().eea_num_modes(self.h))
        self.real_size = int(lib().eea_real_size(self.h))

    def agent_lanes(self, B):
        """eea_batch_agent_lanes: lanes of a wavefront per agent for a plain batch of B agents (64, 8 / 16 / 32, or 0)"""
        return int(lib().eea_batch_agent_lanes(self.h, B))

    def record_count(self, B):
        """eea_batch_record_count: records a launch of B agents writes with rec_per_wavefront set (its wavefronts where agents
        share one, B otherwise)"""
        return int(lib().eea_batch_record_count(self.h, B))

    def close(self):
        if self.h:
            lib().eea_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_target_gaussians(self, mu, sigma):
        import numpy as np
        mu = np.ascontiguousarray(mu, dtype=np.float64).reshape(-1)
        sigma = np.ascontiguousarray(sigma, dtype=np.float64).reshape(-1)
        check(lib().eea_set_target_gaussians(self.h, mu.size // 2, _ptr(mu), _ptr(sigma)))

    def set_target_grid(self, nx, ny, phi_vals, lx, ly, stream=None):
        on_device = 1 if (hasattr(phi_vals, "is_cuda") and phi_vals.is_cuda) else 0
        check(lib().eea_set_target_grid(self.h, nx, ny, _ptr(phi_vals), on_device, lx, ly,
                                        C.c_void_p(stream or 0)))

    def spatial_coeff_rows(self, nx, ny_total, row0, nrows, phi_rows, lx, ly, out_partial, stream=None):
        """partial phi_k of the grid rows [row0, row0+nrows) held in the device tensor phi_rows"""
        check(lib().eea_spatial_coeff_rows(self.h, nx, ny_total, row0, nrows, _ptr(phi_rows), lx, ly,
                                           _ptr(out_partial), C.c_void_p(stream or 0)))

    def set_target_occupancy(self, nx, ny, occ, lx, ly, stream=None):
        """int8 occupancy cells (OccupancyGrid.data layout) -> entropy target -> phi_k, fused on the device"""
        on_device = 1 if (hasattr(occ, "is_cuda") and occ.is_cuda) else 0
        check(lib().eea_set_target_occupancy(self.h, nx, ny, _ptr(occ), on_device, lx, ly,
                                             C.c_void_p(stream or 0)))

    def spatial_coeff_occupancy_rows(self, nx, ny_total, row0, nrows, occ_rows, lx, ly, out_sums, stream=None):
        """un-normalised coefficient sums of the occupancy rows [row0, row0+nrows) (device int8 tensor)"""
        check(lib().eea_spatial_coeff_occupancy_rows(self.h, nx, ny_total, row0, nrows, _ptr(occ_rows), lx, ly,
                                                     _ptr(out_sums), C.c_void_p(stream or 0)))

    def set_phik(self, phik, lx, ly):
        on_device = 1 if (hasattr(phik, "is_cuda") and phik.is_cuda) else 0
        check(lib().eea_set_phik(self.h, _ptr(phik), on_device, lx, ly))

    def set_phik_from_sums(self, sums, lx, ly, stream=None):
        """phi_k = sums / sums[0] on the device, stream-ordered (grid-tiled occupancy target after the all-reduce)"""
        check(lib().eea_set_phik_from_sums(self.h, _ptr(sums), lx, ly, C.c_void_p(stream or 0)))

    def config_domain(self, bounds, stream=None):
        rebuilt = C.c_int(0)
        check(lib().eea_config_domain(self.h, *[float(b) for b in bounds], C.byref(rebuilt),
                                      C.c_void_p(stream or 0)))
        return bool(rebuilt.value)

    def config_domain_async(self, bounds, stream=None):
        """enqueue-only form: returns once the rebuild's launches are on the stream"""
        rebuilt = C.c_int(0)
        check(lib().eea_config_domain_async(self.h, *[float(b) for b in bounds], C.byref(rebuilt),
                                            C.c_void_p(stream or 0)))
        return bool(rebuilt.value)

    def phik(self):
        import numpy as np
        out = np.empty(self.K2)
        check(lib().eea_get_phik(self.h, _ptr(out)))
        return out

    def lamdak(self):
        import numpy as np
        out = np.empty(self.K2)
        check(lib().eea_get_lamdak(self.h, _ptr(out)))
        return out

    def target_grid(self):
        import numpy as np
        nx, ny = C.c_uint(), C.c_uint()
        check(lib().eea_target_grid_size(self.h, C.byref(nx), C.byref(ny)))
        out = np.empty(nx.value * ny.value)
        check(lib().eea_get_target_grid(self.h, _ptr(out)))
        return out, nx.value, ny.value

    def control_batch(self, B, pose, ut, u0, mem_cols=None, n_mem=None, mem_stride=0, traj=None,
                      ck=None, edx=None, bdx=None, rhot=None, status=None, stream=None, ck_shared=None,
                      ck_rec=None, ck_shared_parts=0, n_steps=None, pose_step_stride=0, u0_step_stride=0,
                      rec_ready=None, rec_seq=0, ck_flag=None, ck_flag_seq=0, skip=None,
                      rec_per_wavefront=False):
        """n_steps (ABI 4, eea_control_batch_steps): that many consecutive control() calls per agent in one launch;
        pose / u0 rows per step by the strides (in agents; 0 = the same row every step)."""
        io = BatchIO()
        io.d_ck_shared = _ptr(ck_shared)
        io.d_ck_rec, io.ck_shared_parts = _ptr(ck_rec), ck_shared_parts
        io.d_pose, io.d_ut, io.d_u0 = _ptr(pose), _ptr(ut), _ptr(u0)
        io.d_mem_cols, io.d_n_mem, io.mem_stride = _ptr(mem_cols), _ptr(n_mem), mem_stride
        io.d_traj, io.d_ck, io.d_edx, io.d_bdx = _ptr(traj), _ptr(ck), _ptr(edx), _ptr(bdx)
        io.d_rhot, io.d_status = _ptr(rhot), _ptr(status)
        io.d_rec_ready, io.rec_seq, io.d_ck_flag, io.ck_flag_seq = _ptr(rec_ready), rec_seq, _ptr(ck_flag), ck_flag_seq
        io.d_skip = _ptr(skip)
        io.rec_per_wavefront = 1 if rec_per_wavefront else 0
        if n_steps is None:
            check(lib().eea_control_batch(self.h, B, C.byref(io), C.c_void_p(stream or 0)))
        else:
            check(lib().eea_control_batch_steps(self.h, B, C.byref(io), n_steps, pose_step_stride, u0_step_stride,
                                                C.c_void_p(stream or 0)))

    def prepared_batch(self, B, pose, ut, u0, mem_cols=None, n_mem=None, mem_stride=0, ck=None, ck_shared=None,
                       stream=None, ck_rec=None, ck_shared_parts=0, n_steps=None, pose_step_stride=0, u0_step_stride=0,
                       rec_ready=None, ck_flag=None, status=None):
        """A callable that issues eea_control_batch with these (fixed) device buffers: one ctypes call per pass, the
        eea_batch_io is built once (a pass of 4096 agents takes ~30 us on the device; building the struct from
        tensors every pass costs about as much on the host)."""
        io = BatchIO()
        io.d_ck_shared = _ptr(ck_shared)
        io.d_pose, io.d_ut, io.d_u0 = _ptr(pose), _ptr(ut), _ptr(u0)
        io.d_mem_cols, io.d_n_mem, io.mem_stride = _ptr(mem_cols), _ptr(n_mem), mem_stride
        io.d_ck = _ptr(ck)
        io.d_ck_rec, io.ck_shared_parts = _ptr(ck_rec), ck_shared_parts
        io.d_rec_ready, io.d_ck_flag, io.d_status = _ptr(rec_ready), _ptr(ck_flag), _ptr(status)
        fn, h, ref, st = lib().eea_control_batch, self.h, C.byref(io), C.c_void_p(stream or 0)
        keep = (io, pose, ut, u0, mem_cols, n_mem, ck, ck_shared, ck_rec, rec_ready, ck_flag, status)
        if (rec_ready is not None or ck_flag is not None) and n_steps is not None:
            # (the engine's own answer for eea_control_batch_steps with the device-bound buffers, ergodic_amd.h)
            raise EngineError(ERR_UNSUPPORTED, "n_steps cannot be combined with rec_ready / ck_flag: a step must never wait "
                                               "for an exchange the host enqueues after the launch")
        if rec_ready is not None or ck_flag is not None:
            # device-bound exchange: the sequence numbers change from pass to pass -- call(rec_seq, ck_flag_seq)
            def call_bound(rec_seq=0, ck_flag_seq=0, _keep=keep):
                io.rec_seq, io.ck_flag_seq = rec_seq, ck_flag_seq
                rc = fn(h, B, ref, st)
                if rc != 0:
                    check(rc)
            return call_bound
        if n_steps is not None:   # ABI 4: n_steps receding-horizon steps per launch
            fns = lib().eea_control_batch_steps

            def call_steps(_keep=keep):
                rc = fns(h, B, ref, n_steps, pose_step_stride, u0_step_stride, st)
                if rc != 0:
                    check(rc)
            return call_steps

        def call(_keep=keep):
            rc = fn(h, B, ref, st)
            if rc != 0:
                check(rc)
        return call

    @property
    def ck_record_len(self):
        """reals per sum record (eea_batch_io::d_ck_rec): K^2 + 1 rounded up to even"""
        return int(lib().eea_ck_record_len(self.h))

    def ck_records_sum(self, B, recs, out, stream=None):
        """out[record_len] = sum of the B per-agent records (one launch, fixed order)"""
        check(lib().eea_ck_records_sum(self.h, B, _ptr(recs), _ptr(out), C.c_void_p(stream or 0)))

    def ck_records_sum_bound(self, B, recs, rec_ready, seq, out, flag=None, stream=None):
        """eea_ck_records_sum_bound: the sum polls the agents' ready marks (== seq) itself; flag (optional) = seq when done"""
        check(lib().eea_ck_records_sum_bound(self.h, B, _ptr(recs), _ptr(rec_ready), seq, _ptr(out), _ptr(flag),
                                             C.c_void_p(stream or 0)))

    def publish_record(self, src, pub, flag, seq, stream=None):
        check(lib().eea_publish_record(self.h, _ptr(src), _ptr(pub), _ptr(flag), seq, C.c_void_p(stream or 0)))

    def ck_sum(self, B, ck, sums, stream=None):
        """sums[:K2] = sum over the B agents of ck, sums[K2] = B (device tensors)"""
        check(lib().eea_ck_sum(self.h, B, _ptr(ck), _ptr(sums), C.c_void_p(stream or 0)))

    def debug_phase_timing(self, B, pose, ut, u0, stamps, ck=None, stream=None):
        """A/B library only (EEA_LIB_VARIANT=_ab): tools/phase_timing.py"""
        io = BatchIO()
        io.d_pose, io.d_ut, io.d_u0, io.d_ck = _ptr(pose), _ptr(ut), _ptr(u0), _ptr(ck)
        check(lib().eea_debug_phase_timing(self.h, B, C.byref(io), C.c_void_p(stream or 0), _ptr(stamps)))

    def tick_batch(self, B, pose, ut, follow, count, u, vb, grid, traj, valid, skip, coll, dwa, val_dt, val_horizon,
                   source=None, mem_cols=None, n_mem=None, mem_stride=0, status=None, stream=None, grid_epoch=0):
        """eea_tick_batch: one iteration of Exploration::control's loop body (exploration.hpp:220-279) for B robots.
        coll / dwa: collision_cfg(...) / dwa_cfg(...)"""
        io = BatchIO()
        io.d_pose, io.d_ut = _ptr(pose), _ptr(ut)
        io.d_mem_cols, io.d_n_mem, io.mem_stride, io.d_status = _ptr(mem_cols), _ptr(n_mem), mem_stride, _ptr(status)
        t = TickIO()
        t.d_follow_dwa, t.d_dwa_count, t.d_u, t.d_vb, t.d_grid = _ptr(follow), _ptr(count), _ptr(u), _ptr(vb), _ptr(grid)
        t.d_traj, t.d_valid, t.d_skip, t.d_source = _ptr(traj), _ptr(valid), _ptr(skip), _ptr(source)
        t.val_dt, t.val_horizon, t.grid_epoch = val_dt, val_horizon, grid_epoch
        check(lib().eea_tick_batch(self.h, B, C.byref(io), C.byref(t), C.byref(coll), C.byref(dwa), C.c_void_p(stream or 0)))

    def rollout_batch(self, B, pose, ut, traj, status=None, stream=None):
        check(lib().eea_rollout_batch(self.h, B, _ptr(pose), _ptr(ut), _ptr(traj), _ptr(status),
                                      C.c_void_p(stream or 0)))

    def control(self, bounds, x, mem_cols=None):
        """Single agent, host buffers: mirrors ErgodicControl::control(grid, x)."""
        import numpy as np
        x = np.ascontiguousarray(x, dtype=np.float64)
        u = np.empty(3)
        n_mem, mem = 0, None
        if mem_cols is not None and np.asarray(mem_cols).size:
            mem = np.ascontiguousarray(np.asarray(mem_cols, dtype=np.float64).T)  # (n_mem, 3)
            n_mem = mem.shape[0]
        check(lib().eea_control(self.h, *[float(b) for b in bounds], _ptr(x), _ptr(mem), n_mem,
                                _ptr(u)))
        return u

    def resident_stop(self):
        """eea_resident_stop: the resident single-robot workgroup (OPT_RESIDENT_CONTROL) leaves now; the next control() starts
        another one"""
        check(lib().eea_resident_stop(self.h))

    def opt_traj(self):
        import numpy as np
        out = np.empty((self.T, 3))
        check(lib().eea_opt_traj(self.h, _ptr(out)))
        return out.T.copy()

    def get_ut(self):
        import numpy as np
        out = np.empty((self.T, 3))
        check(lib().eea_get_ut(self.h, _ptr(out)))
        return out.T.copy()

    def set_ut(self, ut):
        import numpy as np
        a = np.ascontiguousarray(np.asarray(ut, dtype=np.float64).T)
        check(lib().eea_set_ut(self.h, _ptr(a)))


def basis_traj_coeff(lx, ly, K, xt, device=0):
    import numpy as np
    xt = np.asarray(xt, dtype=np.float64)
    rows, n = xt.shape
    a = np.ascontiguousarray(xt.T)
    out = np.empty(K * K)
    check(lib().eea_basis_traj_coeff(device, lx, ly, K, _ptr(a), rows, n, _ptr(out)))
    return out


def basis_spatial_coeff(lx, ly, K, phi_vals, grid, device=0):
    import numpy as np
    pv = np.ascontiguousarray(phi_vals, dtype=np.float64)
    g = np.ascontiguousarray(np.asarray(grid, dtype=np.float64).T)
    out = np.empty(K * K)
    check(lib().eea_basis_spatial_coeff(device, lx, ly, K, _ptr(pv), _ptr(g), pv.size, _ptr(out)))
    return out


def rk4_rollout(model, dt, horizon, x0, ut, device=0):
    """RungeKutta::solve for Omni / SimpleCart on the device. ut (3, T) -> xt (3, T)"""
    import numpy as np
    x0 = np.ascontiguousarray(x0, dtype=np.float64)
    a = np.ascontiguousarray(np.asarray(ut, dtype=np.float64).T)
    T = a.shape[0]
    out = np.empty((T, 3))
    check(lib().eea_rk4_rollout(device, model, dt, horizon, _ptr(x0), _ptr(a), _ptr(out)))
    return out.T.copy()


def target_fill(mu, sigma, trans, grid, device=0):
    import numpy as np
    mu = np.ascontiguousarray(mu, dtype=np.float64).reshape(-1)
    sigma = np.ascontiguousarray(sigma, dtype=np.float64).reshape(-1)
    trans = np.ascontiguousarray(trans, dtype=np.float64)
    g = np.ascontiguousarray(np.asarray(grid, dtype=np.float64).T)
    out = np.empty(g.shape[0])
    check(lib().eea_target_fill(device, mu.size // 2, _ptr(mu), _ptr(sigma), _ptr(trans), _ptr(g),
                                g.shape[0], _ptr(out)))
    return out


def make_collision_cfg(xmin, ymin, resolution, xsize, ysize, boundary_radius, search_radius,
                       obstacle_threshold, occupied_threshold):
    return CollisionCfg(xmin, ymin, resolution, xsize, ysize, boundary_radius, search_radius,
                        obstacle_threshold, occupied_threshold)


def collision_check_batch(cfg, grid, pose, hit, device=0, stream=None):
    check(lib().eea_collision_check_batch(device, C.byref(cfg), _ptr(grid), _ptr(pose),
                                          pose.shape[0], _ptr(hit), C.c_void_p(stream or 0)))


def validate_control_batch(cfg, grid, x0, u, dt, horizon, valid, device=0, stream=None):
    check(lib().eea_validate_control_batch(device, C.byref(cfg), _ptr(grid), _ptr(x0), _ptr(u), dt,
                                           horizon, x0.shape[0], _ptr(valid),
                                           C.c_void_p(stream or 0)))


def dwa_control_batch(ccfg, dcfg, grid, x0, vb, u_opt, found, vref=None, xt_ref=None, dt_ref=0.0, device=0,
                      stream=None):
    """DynamicWindow::control for P robots; xt_ref device tensor [P][n_ref][3] or None"""
    n_ref = xt_ref.shape[1] if xt_ref is not None else 0
    check(lib().eea_dwa_control_batch(device, C.byref(ccfg), C.byref(dcfg), _ptr(grid), _ptr(x0), _ptr(vb),
                                      _ptr(vref), _ptr(xt_ref), n_ref, dt_ref, x0.shape[0], _ptr(u_opt),
                                      _ptr(found), C.c_void_p(stream or 0)))


def release_collision_caches():
    """drops the cached ring offsets / inflated-map buffers of the collision, validate and DWA calls"""
    L = lib()
    L.eea_release_collision_caches.restype = None
    L.eea_release_collision_caches()


def set_option(option, value):
    """eea_set_option: process-wide dispatch option (OPT_*)"""
    check(lib().eea_set_option(option, value))


def integrate_twist_batch(x0, u, dt, out=None, normalize_heading=False, stream=None, device=0):
    """eea_integrate_twist_batch (ABI 6): integrate_twist (numerics.hpp:273-297) of P poses on the device; out may be x0"""
    out = x0 if out is None else out
    lib().eea_integrate_twist_batch.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_uint, C.c_void_p, C.c_int, C.c_void_p]
    check(lib().eea_integrate_twist_batch(device, _ptr(x0), _ptr(u), float(dt), int(x0.shape[0]), _ptr(out),
                                          1 if normalize_heading else 0, C.c_void_p(stream or 0)))
    return out


def stream_wait_flag(flag, seq, timeouts=None, stream=None):
    """eea_stream_wait_flag (ABI 6): what follows on `stream` starts once *flag - seq >= 0 -- a one-wavefront gate kernel"""
    check(lib().eea_stream_wait_flag(_ptr(flag), seq, _ptr(timeouts), C.c_void_p(stream or 0)))


def prepared_stream_wait_flag(flag, timeouts=None, stream=None):
    """a callable gate(seq) with the pointers converted once (a pass of 4096 agents takes ~23 us: per-pass host work counts)"""
    fn, f, t, s = lib().eea_stream_wait_flag, _ptr(flag), _ptr(timeouts), C.c_void_p(stream or 0)

    def gate(seq):
        check(fn(f, seq, t, s))
    return gate


def get_option(option):
    return lib().eea_get_option(option)


COMM_ID_BYTES = 128


def comm_set_library(path):
    """binds the collectives to the RCCL at `path` (process-wide, before the first other comm call): eea_comm_set_library"""
    check(lib().eea_comm_set_library(os.fsencode(path)))


def comm_unique_id():
    """128-byte RCCL id created on this rank (rank 0 hands it to the others)"""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    check(lib().eea_comm_get_unique_id(buf))
    return buf.raw


class ConsensusPlan:
    """eea_consensus_plan (ABI 6): `passes_per_launch` consensus passes of a rank -- per pass and agent group one control launch
    (records out, the sum record of pass i - lag in), the record sum and the all-reduce over the ranks -- as ONE replayable
    device graph.  groups: a list of dicts with B, pose, ut, u0 and optionally mem_cols, n_mem, mem_stride, status, skip
    (device tensors; the plan keeps them alive)."""

    def __init__(self, eng, comm, groups, lag=2, passes_per_launch=48):
        self.h = C.c_void_p()
        self._keep = list(groups)
        n = len(groups)
        agents = (C.c_uint * n)(*[int(g["B"]) for g in groups])
        ios = (BatchIO * n)()
        for io, g in zip(ios, groups):
            io.d_pose, io.d_ut, io.d_u0 = _ptr(g["pose"]), _ptr(g["ut"]), _ptr(g["u0"])
            io.d_mem_cols, io.d_n_mem, io.mem_stride = _ptr(g.get("mem_cols")), _ptr(g.get("n_mem")), int(g.get("mem_stride", 0) or 0)
            io.d_status, io.d_skip = _ptr(g.get("status")), _ptr(g.get("skip"))
        d = ConsensusDesc(n, agents, ios, lag, passes_per_launch)
        check(lib().eea_consensus_plan_create(eng.h, comm.h, C.byref(d), C.byref(self.h)))
        passes, last = C.c_uint(), C.c_void_p()
        check(lib().eea_consensus_plan_info(self.h, C.byref(passes), C.byref(last)))
        self.passes_per_launch, self.d_last_sum = passes.value, last.value
        self._launch = lib().eea_consensus_plan_launch

    def launch(self, stream=None):
        check(self._launch(self.h, C.c_void_p(stream or 0)))

    def last_sum(self, eng):
        """the sum record the last pass of a launch leaves, as a host array [eea_ck_record_len] (synchronise first)"""
        import numpy as np
        n = eng.ck_record_len
        out = np.empty(n, dtype=np.float32 if eng.real_size == 4 else np.float64)
        hip = C.CDLL("libamdhip64.so")   # (already mapped: the library links it)
        hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        if hip.hipMemcpy(out.ctypes.data, self.d_last_sum, out.nbytes, 2) != 0:   # hipMemcpyDeviceToHost
            raise EngineError(ERR_HIP, "hipMemcpy of the plan's sum record failed")
        return out

    def close(self):
        if self.h:
            lib().eea_consensus_plan_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Comm:
    """Communicator of the agent-batch exchange steps (RCCL over xGMI; ncclComm behind the C ABI).
    uid=None with nranks == 1: local communicator without RCCL."""

    def __init__(self, device, nranks, rank, uid=None):
        self.h = C.c_void_p()
        idbuf = C.create_string_buffer(uid, COMM_ID_BYTES) if uid is not None else None
        check(lib().eea_comm_create(device, nranks, rank, idbuf, C.byref(self.h)))
        self.nranks, self.rank = nranks, rank

    def close(self):
        if self.h:
            lib().eea_comm_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def library_nranks(self):
        """ncclCommCount of the communicator behind the C ABI (0: local communicator, -1: the library cannot tell)"""
        lib().eea_comm_library_nranks.argtypes = [C.c_void_p]
        return int(lib().eea_comm_library_nranks(self.h))

    def allgather_ck(self, eng, B_local, ck_local, ck_all, stream=None):
        check(lib().eea_comm_allgather_ck(eng.h, self.h, B_local, _ptr(ck_local), _ptr(ck_all),
                                          C.c_void_p(stream or 0)))

    def consensus_ck(self, eng, B_local, ck_local, ck_shared, stream=None):
        check(lib().eea_comm_consensus_ck(eng.h, self.h, B_local, _ptr(ck_local), _ptr(ck_shared),
                                          C.c_void_p(stream or 0)))

    def consensus_ck_async(self, eng, B_local, ck_local, ck_shared, compute_stream, slot):
        check(lib().eea_comm_consensus_ck_async(eng.h, self.h, B_local, _ptr(ck_local), _ptr(ck_shared),
                                                C.c_void_p(compute_stream or 0), slot))

    def allgather_ck_async(self, eng, B_local, ck_local, ck_all, compute_stream, slot):
        check(lib().eea_comm_allgather_ck_async(eng.h, self.h, B_local, _ptr(ck_local), _ptr(ck_all),
                                                C.c_void_p(compute_stream or 0), slot))

    def wait(self, slot, stream=None):
        check(lib().eea_comm_wait(self.h, slot, C.c_void_p(stream or 0)))

    def records_exchange_async(self, eng, B_local, ck_rec, out_sum, group_streams, slot):
        """eea_comm_records_exchange_async: record sum + all-reduce on the communicator's stream, after every group stream"""
        arr = (C.c_void_p * len(group_streams))(*[C.c_void_p(s or 0) for s in group_streams])
        check(lib().eea_comm_records_exchange_async(eng.h, self.h, B_local, _ptr(ck_rec), _ptr(out_sum), arr,
                                                    len(group_streams), slot))

    def prepared_records_exchange(self, eng, B_local, ck_rec, out_sum, group_streams, slot):
        """the same as a callable with the argument marshalling done once (bench.py's per-pass call)"""
        arr = (C.c_void_p * len(group_streams))(*[C.c_void_p(s or 0) for s in group_streams])
        fn, args = lib().eea_comm_records_exchange_async, (eng.h, self.h, B_local, _ptr(ck_rec), _ptr(out_sum), arr,
                                                           len(group_streams), slot)
        keep = (ck_rec, out_sum, arr)

        def call(_keep=keep):
            rc = fn(*args)
            if rc != 0:
                check(rc)
        return call

    def records_exchange_bound(self, eng, B_local, ck_rec, rec_ready, seq, out_sum, flag, slot):
        """eea_comm_records_exchange_bound: the device-bound exchange of one pass (no host or stream waits anywhere)"""
        check(lib().eea_comm_records_exchange_bound(eng.h, self.h, B_local, _ptr(ck_rec), _ptr(rec_ready), seq,
                                                    _ptr(out_sum), _ptr(flag), slot))

    def prepared_records_exchange_bound(self, eng, B_local, ck_rec, rec_ready, out_sum, flag, slot):
        """the same as a callable call(seq) with the argument marshalling done once"""
        fn, h, eh = lib().eea_comm_records_exchange_bound, self.h, eng.h
        a = (_ptr(ck_rec), _ptr(rec_ready), _ptr(out_sum), _ptr(flag))
        keep = (ck_rec, rec_ready, out_sum, flag)

        def call(seq, _keep=keep):
            rc = fn(eh, h, B_local, a[0], a[1], seq, a[2], a[3], slot)
            if rc != 0:
                check(rc)
        return call

    def prepared_wait(self, slot, stream):
        fn, args = lib().eea_comm_wait, (self.h, slot, C.c_void_p(stream or 0))

        def call():
            rc = fn(*args)
            if rc != 0:
                check(rc)
        return call

    def allreduce_sum_async(self, eng, buf, n, compute_stream, slot):
        check(lib().eea_comm_allreduce_sum_async(eng.h, self.h, _ptr(buf), n, C.c_void_p(compute_stream or 0), slot))

    def allreduce_sum(self, eng, buf, n, stream=None):
        check(lib().eea_comm_allreduce_sum(eng.h, self.h, _ptr(buf), n, C.c_void_p(stream or 0)))
