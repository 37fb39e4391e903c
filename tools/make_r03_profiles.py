#!/usr/bin/env python3
"""Copies what tools/r03_evidence.sh measured (gpurun_out/r03_evidence, gpurun_out/prof_r03_*) into profiles/ and
writes the small JSON records bench.py reads back (profiles/r03_bench_profile.json, r03_control_pmc.json,
r03_phik_pmc.json)."""
import json
import os
import re
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EV = os.path.join(ROOT, "gpurun_out", "r03_evidence")
PR = os.path.join(ROOT, "profiles")


def cp(src, dst):
    if os.path.exists(os.path.join(EV, src)):
        shutil.copy(os.path.join(EV, src), os.path.join(PR, dst))


def main():
    # bench_final.json: the default command once more AFTER this script ran on the box (tools/r03_evidence.sh), so that
    # the roofline.*_profiled fields it reads back from profiles/ come from the same box and build as the line itself
    name = "bench_final.json" if os.path.exists(os.path.join(EV, "bench_final.json")) else "bench.json"
    line = open(os.path.join(EV, name)).read().strip().splitlines()[-1]
    bench = json.loads(line)
    with open(os.path.join(PR, "r03_bench.json"), "w") as f:
        f.write(line + "\n")
    for g in ("g2", "g1"):
        cp("%s_summary.txt" % g, "r03_%s_summary.txt" % g)
        cp("%s_summary.json" % g, "r03_%s_summary.json" % g)
        cp("%s_kernel_stats.csv" % g, "r03_%s_kernel_stats.csv" % g)
    for src, dst in (("phase_timing.txt", "r03_phase_timing.txt"), ("phik_pmc.txt", "r03_phik_pmc.txt"),
                     ("parity_report.txt", "r03_parity_report.txt"), ("analytic_checks.txt", "r03_analytic_checks.txt"),
                     ("ck_cost.txt", "r03_exchange_cost.txt"), ("config_sweep.txt", "r03_config_sweep.txt")):
        cp(src, dst)
    g2 = json.load(open(os.path.join(EV, "g2_summary.json")))
    g1 = json.load(open(os.path.join(EV, "g1_summary.json")))
    clock = None
    m = re.search(r"shader clock while the wavefronts ran: ([0-9.]+) GHz", open(os.path.join(EV, "phase_timing.txt")).read())
    if m:
        clock = float(m.group(1))
    tr = g2["timed_region"]
    rec = {"command": "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 --passes-per-step 1000 "
                      "(default shape: clock spin-up on, two agent groups; tools/profile_r.sh)",
           "kernel": "control_wave_kernel_lean<double, SimpleCart, 10>", "agents": bench["config"]["agents_per_gpu"],
           "T": bench["config"]["horizon_steps"], "K": bench["config"]["num_basis"], "precision": bench["dtype"],
           "agents_per_launch": tr["agents_per_launch"], "concurrent_launches": tr["concurrent_launches"],
           "kernel_avg_us_profiled": tr["kernel_avg_us_timed_region"],
           "dispatches_averaged": tr["dispatches_timed_region"],
           "pass_period_us_from_trace": tr["pass_period_us_from_trace"],
           "bench_ms_per_pass_in_the_profiled_run": tr["bench_ms_per_pass_same_run"],
           "frac_of_78.6TF_from_kernel_avg": tr["frac_of_78.6_from_kernel_avg"],
           "one_launch_per_pass": {"kernel_avg_us_profiled": g1["timed_region"]["kernel_avg_us_timed_region"],
                                   "pass_period_us_from_trace": g1["timed_region"]["pass_period_us_from_trace"],
                                   "bench_ms_per_pass_in_the_profiled_run": g1["timed_region"]["bench_ms_per_pass_same_run"]},
           "effective_clock_ghz": clock,
           "effective_clock_method": "phase stamps of the A/B library after 1000 passes of load: shader cycles (s_memtime) "
                                     "/ lifetime (s_memrealtime, 100 MHz) averaged over the 4096 wavefronts of one launch "
                                     "(profiles/r03_phase_timing.txt)",
           "source": "profiles/r03_g2_summary.txt, profiles/r03_g1_summary.txt"}
    json.dump(rec, open(os.path.join(PR, "r03_bench_profile.json"), "w"), indent=1)
    # HBM traffic of one launch of 4096 agents (the one-launch-per-pass profile), as bench.py scales it
    if "hbm_bytes_per_launch" in g1:
        B, T = 4096, bench["config"]["horizon_steps"]
        pmc = {"agents": B, "T": T, "K": bench["config"]["num_basis"], "precision": bench["dtype"],
               "kernel": "control_wave_kernel_lean<double, SimpleCart, 10> (r03: 4x4-block contraction, 120 registers), one launch per pass",
               "fetch_size_kib": g1["pmc_mean_per_dispatch"].get("FETCH_SIZE"),
               "write_size_kib": g1["pmc_mean_per_dispatch"].get("WRITE_SIZE"),
               "hbm_read_bytes_x2_corrected": g1["hbm_read_bytes_x2_corrected"], "hbm_write_bytes": g1["hbm_write_bytes_raw"],
               "hbm_bytes_per_launch": g1["hbm_bytes_per_launch"], "algorithmic_bytes_per_launch": 8 * (3 + 6 * T + 3) * B,
               "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/profile_r.sh r03_g1 --agent-groups 1; "
                       "profiles/r03_g1_summary.json); FETCH_SIZE doubled (gfx950 correction, MI355X_MICROARCH.md)"}
        json.dump(pmc, open(os.path.join(PR, "r03_control_pmc.json"), "w"), indent=1)
    # phi_k streaming kernel at the bench's grid
    txt = open(os.path.join(EV, "phik_pmc.txt")).read()
    m = re.search(r"(\d+):(\d+):f64\s+FETCH_SIZE (\d+) KiB raw", txt)
    if m:
        n, K, kib = int(m.group(1)), int(m.group(2)), float(m.group(3))
        busy = re.search(r"SQ_VALU_MFMA_BUSY_CYCLES ([0-9.e+]+)", txt)
        gui = re.search(r"GRBM_GUI_ACTIVE ([0-9.e+]+)", txt)
        rec = {"grid": n, "K": K, "precision": "f64", "kernel": "spatial_stream_kernel<double, 1, double>",
               "fetch_size_kib_raw": kib, "hbm_read_bytes_x2_corrected": 2.0 * kib * 1024.0, "algorithmic_bytes": n * n * 8,
               "ratio": 2.0 * kib * 1024.0 / (n * n * 8),
               "SQ_VALU_MFMA_BUSY_CYCLES": float(busy.group(1)) if busy else None,
               "GRBM_GUI_ACTIVE": float(gui.group(1)) if gui else None,
               "note": "rocprofv3 --pmc FETCH_SIZE (own pass) and MFMA busy / GRBM cycles (own pass), tools/phik_pmc.sh with "
                       "PHIK_CASES=16384:10:f64; FETCH_SIZE doubled (gfx950 correction); mean of 5 dispatches"}
        json.dump(rec, open(os.path.join(PR, "r03_phik_pmc.json"), "w"), indent=1)
    print("profiles/ updated from", EV)


if __name__ == "__main__":
    main()
