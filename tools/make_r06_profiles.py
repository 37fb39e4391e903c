#!/usr/bin/env python3
"""Copies what tools/r06_evidence.sh measured (gpurun_out/r06_evidence) into profiles/ and writes the small JSON records
bench.py reads back (profiles/r06_bench_profile.json, r06_control_pmc.json)."""
import json
import os
import re
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EV = os.path.join(ROOT, "gpurun_out", "r06_evidence")
PR = os.path.join(ROOT, "profiles")


def cp(src, dst):
    if os.path.exists(os.path.join(EV, src)):
        shutil.copy(os.path.join(EV, src), os.path.join(PR, dst))


def main():
    name = "bench_final.json" if os.path.exists(os.path.join(EV, "bench_final.json")) else "bench.json"
    line = open(os.path.join(EV, name)).read().strip().splitlines()[-1]
    bench = json.loads(line)
    with open(os.path.join(PR, "r06_bench.json"), "w") as f:
        f.write(line + "\n")
    for src, dst in (("bench_detail.json", "r06_bench_detail.json"), ("exchange_modes.txt", "r06_exchange_modes.txt"),
                     ("pack_summary.txt", "r06_pack_summary.txt"), ("pack_profile.json", "r06_pack_profile.json"),
                     ("pack_pmc.txt", "r06_pack_pmc.txt"), ("pack_points.txt", "r06_pack_points.txt"),
                     ("pack_c2_kernel_stats.csv", "r06_pack_c2_kernel_stats.csv"), ("pack_c3_kernel_stats.csv", "r06_pack_c3_kernel_stats.csv"),
                     ("packed_consensus.txt", "r06_packed_consensus.txt")):
        cp(src, dst)
    for g in ("spl50", "spl1", "k20_f32"):
        cp("%s_summary.txt" % g, "r06_%s_summary.txt" % g)
        cp("%s_summary.json" % g, "r06_%s_summary.json" % g)
        cp("%s_kernel_stats.csv" % g, "r06_%s_kernel_stats.csv" % g)
    for src, dst in (("phase_timing.txt", "r06_phase_timing.txt"), ("parity_report.txt", "r06_parity_report.txt"),
                     ("analytic_checks.txt", "r06_analytic_checks.txt"), ("exchange_cost.txt", "r06_exchange_cost.txt"),
                     ("rebuild_kernels.txt", "r06_rebuild_kernels.txt")):
        cp(src, dst)
    g50 = json.load(open(os.path.join(EV, "spl50_summary.json")))
    g1 = json.load(open(os.path.join(EV, "spl1_summary.json")))
    clock = None
    try:
        m = re.search(r"shader clock while the wavefronts ran: ([0-9.]+) GHz", open(os.path.join(EV, "phase_timing.txt")).read())
        if m:
            clock = float(m.group(1))
    except OSError:
        pass
    tr = g50["timed_region"]
    rec = {"command": "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 --passes-per-step 2000 "
                      "(default shape: clock spin-up on, two agent groups, 50 receding-horizon steps per launch; tools/profile_r.sh)",
           "kernel": "control_wave_kernel<double, SimpleCart, 10, STAGES = false>", "agents": bench["config"]["agents_per_gpu"],
           "T": bench["config"]["horizon_steps"], "K": bench["config"]["num_basis"], "precision": bench["dtype"],
           "steps_per_launch": tr.get("steps_per_launch", 1),
           "agents_per_launch": tr["agents_per_launch"], "concurrent_launches": tr["concurrent_launches"],
           "kernel_avg_us_profiled": tr["kernel_avg_us_timed_region"],
           "kernel_avg_us_per_pass_profiled": tr["pass_us_from_kernel_avg"],
           "dispatches_averaged": tr["dispatches_timed_region"],
           "pass_period_us_from_trace": tr["pass_period_us_from_trace"],
           "bench_ms_per_pass_in_the_profiled_run": tr["bench_ms_per_pass_same_run"],
           "frac_of_78.6TF_from_kernel_avg": tr["frac_of_78.6_from_kernel_avg"],
           "one_launch_per_pass": {"kernel_avg_us_profiled": g1["timed_region"]["kernel_avg_us_timed_region"],
                                   "pass_period_us_from_trace": g1["timed_region"]["pass_period_us_from_trace"],
                                   "bench_ms_per_pass_in_the_profiled_run": g1["timed_region"]["bench_ms_per_pass_same_run"]},
           "effective_clock_ghz": clock,
           "source": "profiles/r06_spl50_summary.txt, profiles/r06_spl1_summary.txt"}
    json.dump(rec, open(os.path.join(PR, "r06_bench_profile.json"), "w"), indent=1)
    # HBM traffic per launch of the profiled shape (PMC passes of the same command: 2048 agents x 50 steps per dispatch)
    for tag, src in (("", g50), ("_spl1", g1)):
        if "hbm_bytes_per_launch" not in src:
            continue
        T = bench["config"]["horizon_steps"]
        tr_ = src["timed_region"]
        spl = tr_.get("steps_per_launch", 1)
        apl = tr_["agents_per_launch"]
        pmc = {"agents_per_launch": apl, "steps_per_launch": spl, "T": T, "K": bench["config"]["num_basis"],
               "precision": bench["dtype"], "agents": bench["config"]["agents_per_gpu"],
               "kernel": "control_wave_kernel<double, SimpleCart, 10, STAGES = false> (r06), %d agents x %d receding-horizon steps per launch" % (apl, spl),
               "fetch_size_kib": src["pmc_mean_per_dispatch"].get("FETCH_SIZE"),
               "write_size_kib": src["pmc_mean_per_dispatch"].get("WRITE_SIZE"),
               "hbm_read_bytes_x2_corrected": src["hbm_read_bytes_x2_corrected"], "hbm_write_bytes": src["hbm_write_bytes_raw"],
               "hbm_bytes_per_launch": src["hbm_bytes_per_launch"],
               "algorithmic_bytes_per_launch": 8 * (3 + 6 * T + 3) * apl * spl,
               "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of the profiled command shape (tools/profile_r.sh); "
                       "FETCH_SIZE doubled (gfx950 correction, MI355X_MICROARCH.md).  With several steps per launch the controls "
                       "a step stores are read back by the next one from L2: the fetch side falls below the algorithmic bytes"}
        json.dump(pmc, open(os.path.join(PR, "r06_control_pmc%s.json" % tag), "w"), indent=1)
    # phi_k streaming kernel: HBM traffic of this round's build (tools/phik_pmc.sh, PHIK_CASES=16384:10:f64)
    try:
        txt = open(os.path.join(EV, "phik_pmc.txt")).read()
        shutil.copy(os.path.join(EV, "phik_pmc.txt"), os.path.join(PR, "r06_phik_pmc.txt"))
        m = re.search(r"FETCH_SIZE (\d+) KiB raw", txt)
        busy = re.search(r"SQ_VALU_MFMA_BUSY_CYCLES ([0-9.e+]+)", txt)
        grbm = re.search(r"GRBM_GUI_ACTIVE ([0-9.e+]+)", txt)
        if m:
            raw = float(m.group(1))
            rec = {"grid": 16384, "K": 10, "precision": "f64", "kernel": "spatial_stream_kernel<double, 1, double>",
                   "fetch_size_kib_raw": raw, "hbm_read_bytes_x2_corrected": 2.0 * raw * 1024.0, "algorithmic_bytes": 16384 * 16384 * 8,
                   "ratio": 2.0 * raw * 1024.0 / (16384 * 16384 * 8),
                   "SQ_VALU_MFMA_BUSY_CYCLES": float(busy.group(1)) if busy else None,
                   "GRBM_GUI_ACTIVE": float(grbm.group(1)) if grbm else None,
                   "note": "rocprofv3 --pmc FETCH_SIZE (own pass) and MFMA busy / GRBM cycles (own pass), tools/phik_pmc.sh with "
                           "PHIK_CASES=16384:10:f64 on the round-6 build; FETCH_SIZE doubled (gfx950 correction); mean of 5 dispatches"}
            json.dump(rec, open(os.path.join(PR, "r06_phik_pmc.json"), "w"), indent=1)
    except OSError:
        pass
    # instruction counts of the metric kernel from the SQ counters (per wavefront = per agent and pass)
    try:
        pm = g1.get("pmc_mean_per_dispatch", {})
        waves = pm.get("SQ_WAVES")
        if waves:
            spl1 = g1["timed_region"].get("steps_per_launch", 1)
            rec = {"source": "profiles/r06_spl1_summary.txt (rocprofv3 --pmc SQ_* passes, one step per launch)",
                   "T": bench["config"]["horizon_steps"], "K": bench["config"]["num_basis"], "precision": bench["dtype"],
                   "valu_insts_per_wave_incl_matrix": pm.get("SQ_INSTS_VALU") / waves / spl1,
                   "matrix_insts_per_wave": (pm.get("SQ_INSTS_VALU_MFMA_MOPS_F64") or 0) / waves / spl1 if pm.get("SQ_INSTS_VALU_MFMA_MOPS_F64") else None,
                   "lds_insts_per_wave": pm.get("SQ_INSTS_LDS") / waves / spl1 if pm.get("SQ_INSTS_LDS") else None,
                   "salu_insts_per_wave": pm.get("SQ_INSTS_SALU") / waves / spl1 if pm.get("SQ_INSTS_SALU") else None,
                   "wait_inst_any_over_wave_cycles": (pm.get("SQ_WAIT_INST_ANY") / pm.get("SQ_WAVE_CYCLES")) if pm.get("SQ_WAVE_CYCLES") else None}
            json.dump(rec, open(os.path.join(PR, "r06_isa_counts.json"), "w"), indent=1)
    except Exception as exc:  # noqa: BLE001
        print("isa counts:", exc)
    print("profiles/ updated from", EV)


if __name__ == "__main__":
    main()
