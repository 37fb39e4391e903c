"""CPU-only checks of the drop-in boundary: the C-ABI library builds for gfx950, loads without a
GPU, exports every entry point include/ergodic_amd.h declares, and refuses to compute without a
device (no CPU fallback)."""
import ctypes as C
import os

import numpy as np
import pytest

from ergodic_exploration_amd import capi


def test_library_exports_every_declared_symbol():
    assert os.path.exists(capi.LIB_PATH), "run __graft_entry__.build() first"
    L = C.CDLL(capi.LIB_PATH)
    syms = capi.declared_symbols()
    assert len(syms) >= 25
    missing = [s for s in syms if not hasattr(L, s)]
    assert missing == []
    assert capi.lib().eea_abi_version() == 6


def test_library_exports_only_the_documented_abi():
    # the product library carries no A/B kernels or diagnostics: its dynamic symbol table is exactly the
    # set of entry points the header declares (csrc/exports.map; the superseded kernels live in tools/ab/)
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", capi.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted(line.split()[-1] for line in out.splitlines() if line.strip())
    assert exported == capi.declared_symbols()


def test_struct_layouts_match_header():
    # sizes the C compiler gives the ABI structs (kept in sync with ergodic_amd.h by hand)
    assert C.sizeof(capi.Config) == 3 * 4 + 4 + 4 * 8 + 8 + 15 * 8  # ints, pad, doubles, K+pad, arrays
    assert C.sizeof(capi.BatchIO) == 15 * 8 + 4 * 8 + 8 + 8  # ABI 3 (15 slots) + d_rec_ready, rec_seq, d_ck_flag, ck_flag_seq (ABI 4) + d_skip (ABI 5)
    # + rec_per_wavefront and its tail padding (ABI 6)
    assert C.sizeof(capi.CollisionCfg) == 3 * 8 + 2 * 4 + 4 * 8


def test_struct_layouts_against_the_c_compiler(tmp_path):
    """sizeof / offsetof of every ABI struct as gcc lays them out from include/ergodic_amd.h, against the ctypes mirrors"""
    import subprocess
    structs = {"eea_config": capi.Config, "eea_batch_io": capi.BatchIO, "eea_collision_cfg": capi.CollisionCfg,
               "eea_dwa_cfg": capi.DwaCfg, "eea_tick_io": capi.TickIO}
    lines = []
    for cname, cls in structs.items():
        lines.append('printf("%s %%zu\\n", sizeof(%s));' % (cname, cname))
        for fname, _ in cls._fields_:
            lines.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (cname, fname, cname, fname))
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "ergodic_amd.h"\nint main(void) {\n%s\nreturn 0; }\n'
                   % "\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.dirname(capi.HEADER_PATH), str(src), "-o", str(exe)], check=True)
    got = dict(line.split() for line in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.splitlines())
    for cname, cls in structs.items():
        assert int(got[cname]) == C.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert int(got["%s.%s" % (cname, fname)]) == getattr(cls, fname).offset, (cname, fname)


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    cfg = capi.make_config(capi.MODEL_OMNI, 0.1, 2.0, 0.1, 1.0, 10, np.eye(3), [-1] * 3, [1] * 3)
    with pytest.raises(capi.EngineError) as ei:
        capi.Engine(cfg)
    assert ei.value.status == capi.ERR_HIP


def test_argument_errors_do_not_need_a_device():
    L = capi.lib()
    assert L.eea_create(None, None) == capi.ERR_INVALID_ARGUMENT
    h = C.c_void_p()
    bad = capi.make_config(capi.MODEL_OMNI, 0.1, 0.1, 0.1, 1.0, 10, np.eye(3), [-1] * 3, [1] * 3)
    assert L.eea_create(C.byref(bad), C.byref(h)) == capi.ERR_INVALID_ARGUMENT  # steps == 1
    assert b"two steps" in L.eea_last_error()
    bad = capi.make_config(3, 0.1, 1.0, 0.1, 1.0, 10, np.eye(3), [-1] * 3, [1] * 3)
    assert L.eea_create(C.byref(bad), C.byref(h)) == capi.ERR_INVALID_ARGUMENT  # Mecanum
    bad = capi.make_config(capi.MODEL_OMNI, 0.1, 1.0, 0.1, 1.0, 64, np.eye(3), [-1] * 3, [1] * 3)
    assert L.eea_create(C.byref(bad), C.byref(h)) == capi.ERR_UNSUPPORTED
    # exchange steps: argument checks come before any device / RCCL call
    assert L.eea_comm_create(0, 2, 5, None, C.byref(h)) == capi.ERR_INVALID_ARGUMENT
    assert L.eea_comm_create(0, 2, 0, None, C.byref(h)) == capi.ERR_INVALID_ARGUMENT  # nranks > 1 needs an id
    assert L.eea_comm_get_unique_id(None) == capi.ERR_INVALID_ARGUMENT
    assert L.eea_comm_nranks(None) == 1 and L.eea_comm_rank(None) == 0


def test_options_need_no_device_and_reject_bad_values():
    """eea_set_option / eea_get_option (the library reads no environment variable): defaults, round trip, range checks"""
    assert [capi.get_option(o) for o in range(4)] == [0, 0, 0, 1]
    capi.set_option(capi.OPT_COLLISION_IMPL, 2)
    assert capi.get_option(capi.OPT_COLLISION_IMPL) == 2
    capi.set_option(capi.OPT_COLLISION_IMPL, 0)
    for opt, bad in ((capi.OPT_CONTROL_KERNEL, 2), (capi.OPT_WORKGROUP_THREADS, 96), (capi.OPT_COLLISION_IMPL, 3),
                     (capi.OPT_MAILBOX_POLL, -1), (17, 0)):
        with pytest.raises(capi.EngineError):
            capi.set_option(opt, bad)
    assert capi.get_option(99) == 0


def test_library_reads_no_environment_variable():
    """the knobs of ABI 2 (EEA_CONTROL_PATH, EEA_BLOCK, ...) are gone: no getenv in the product library"""
    import subprocess
    out = subprocess.run(["nm", "-D", "--undefined-only", capi.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in out


def test_timed_control_instances_leave_room_for_the_record_sum():
    """The fp64 K <= 10 instances of the wavefront kernel -- what bench.py times -- must stay at <= 120 registers without
    scratch: four wavefronts per SIMD then leave 32 registers (allocation granule 8) for the record sum's wavefronts
    BESIDE a fully resident control kernel, and the record sum itself must fit into those 32.  Read from
    the gfx950 code objects of the build (tools/kernel_resources.py); until round 4 a second, register-capped compilation
    of the kernel text guaranteed this."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import subprocess
    import kernel_resources as kr
    build = os.path.join(os.path.dirname(capi.LIB_PATH), "..", "csrc", "build")
    def demangled(path):
        out = {}
        for k in kr.kernels(path):
            if "vgpr_count" in k:
                name = subprocess.run(["c++filt", k["name"]], capture_output=True, text=True).stdout.strip()
                out[name] = k
        return out
    wave = demangled(os.path.join(build, "control_wave_kernel.o"))
    seen = 0
    for name, k in wave.items():
        if "control_wave_kernel<double, " in name and (", 10, " in name or ", 5, " in name):
            resident = ", true>(" in name   # the resident single-robot wavefront: alone on its SIMD, no register bar
            assert (resident or int(k["vgpr_count"]) <= 120) and int(k["private_segment_fixed_size"]) == 0, (name, k)
            seen += 1 if not resident else 100
    assert seen == 408   # 2 models x K in {5, 10} x stage outputs on / off, + 2 x 2 resident instances
    sums = [k for n, k in demangled(os.path.join(build, "control_kernel.o")).items() if "ck_records_sum_kernel" in n]
    assert len(sums) == 2 and all(int(k["vgpr_count"]) <= 32 and int(k["group_segment_fixed_size"]) == 0 for k in sums), sums


def test_packed_timed_instances_fit_four_wavefronts_per_simd():
    """Round 6: the timed instances of the several-agents-per-wavefront kernel (no stage outputs) that are compiled for FOUR
    wavefronts per SIMD take <= 128 registers and no scratch: K = 5 at every group size; K = 10 at 16 lanes per agent (yaml's
    T = 50: round 5 had 162-164 registers, three per SIMD) and at 8 lanes per agent with <= 3 steps per lane (configs[1]'s
    T = 20).  What keeps them there: lambda_k / phi_k read where D is formed (not preloaded), one accumulator set for the four
    agents of a 16-lane wavefront, three-element per-step arrays for horizons of <= 3 steps per lane
    (csrc/control_pack_impl.hpp).  No instance uses scratch but one, which parks one value (below)."""
    import re
    import subprocess
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import kernel_resources as kr
    build = os.path.join(os.path.dirname(capi.LIB_PATH), "..", "csrc", "build")
    seen = 0
    for k in kr.kernels(os.path.join(build, "control_pack_kernel.o")):
        if "vgpr_count" not in k:
            continue
        name = subprocess.run(["c++filt", k["name"]], capture_output=True, text=True).stdout.strip()
        m = re.search(r"control_pack_kernel<(\d+), (\d+), (true|false), (\d+), (\d+), (\d+)>", name)
        if not m:
            continue
        KC, stages, L, SM = int(m.group(2)), m.group(3) == "true", int(m.group(4)), int(m.group(6))
        # (the 8-lane K = 10 instance at <= 3 steps per lane sits exactly at 128 registers: with the per-wavefront sum records in
        # the kernel the allocator parks ONE per-step sine in scratch across the contraction -- one store after the rollout, one
        # load in the tail, per wavefront; measured: no change of configs[1]'s pass time, profiles/r06_ablation.txt item 12)
        parked = 12 if (KC == 10 and L == 8 and SM == 3 and not stages) else 0
        assert int(k["private_segment_fixed_size"]) <= parked, (name, k)
        four = not stages and (KC == 5 or L == 16 or (L == 8 and SM == 3))
        if four:
            assert int(k["vgpr_count"]) <= 128, (name, k)
            seen += 1
        elif KC == 10:
            assert int(k["vgpr_count"]) <= 168, (name, k)   # three per SIMD
    assert seen == 2 * (3 * 2 + 2 + 1)   # 2 models x (K = 5: 3 group sizes x 2 SM; K = 10: L = 16 x 2 SM, L = 8 at SM = 3)
