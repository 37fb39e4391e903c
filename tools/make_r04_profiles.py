#!/usr/bin/env python3
"""Copies what tools/r04_evidence.sh measured (gpurun_out/r04_evidence) into profiles/ and writes the small JSON records
bench.py reads back (profiles/r04_bench_profile.json, r04_control_pmc.json)."""
import json
import os
import re
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EV = os.path.join(ROOT, "gpurun_out", "r04_evidence")
PR = os.path.join(ROOT, "profiles")


def cp(src, dst):
    if os.path.exists(os.path.join(EV, src)):
        shutil.copy(os.path.join(EV, src), os.path.join(PR, dst))


def main():
    name = "bench_final.json" if os.path.exists(os.path.join(EV, "bench_final.json")) else "bench.json"
    line = open(os.path.join(EV, name)).read().strip().splitlines()[-1]
    bench = json.loads(line)
    with open(os.path.join(PR, "r04_bench.json"), "w") as f:
        f.write(line + "\n")
    for g in ("spl50", "spl1", "k20_f32"):
        cp("%s_summary.txt" % g, "r04_%s_summary.txt" % g)
        cp("%s_summary.json" % g, "r04_%s_summary.json" % g)
        cp("%s_kernel_stats.csv" % g, "r04_%s_kernel_stats.csv" % g)
    for src, dst in (("phase_timing.txt", "r04_phase_timing.txt"), ("parity_report.txt", "r04_parity_report.txt"),
                     ("analytic_checks.txt", "r04_analytic_checks.txt"), ("exchange_cost.txt", "r04_exchange_cost.txt"),
                     ("rebuild_kernels.txt", "r04_rebuild_kernels.txt")):
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
           "source": "profiles/r04_spl50_summary.txt, profiles/r04_spl1_summary.txt"}
    json.dump(rec, open(os.path.join(PR, "r04_bench_profile.json"), "w"), indent=1)
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
               "kernel": "control_wave_kernel<double, SimpleCart, 10, STAGES = false> (r04), %d agents x %d receding-horizon steps per launch" % (apl, spl),
               "fetch_size_kib": src["pmc_mean_per_dispatch"].get("FETCH_SIZE"),
               "write_size_kib": src["pmc_mean_per_dispatch"].get("WRITE_SIZE"),
               "hbm_read_bytes_x2_corrected": src["hbm_read_bytes_x2_corrected"], "hbm_write_bytes": src["hbm_write_bytes_raw"],
               "hbm_bytes_per_launch": src["hbm_bytes_per_launch"],
               "algorithmic_bytes_per_launch": 8 * (3 + 6 * T + 3) * apl * spl,
               "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of the profiled command shape (tools/profile_r.sh); "
                       "FETCH_SIZE doubled (gfx950 correction, MI355X_MICROARCH.md).  With several steps per launch the controls "
                       "a step stores are read back by the next one from L2: the fetch side falls below the algorithmic bytes"}
        json.dump(pmc, open(os.path.join(PR, "r04_control_pmc%s.json" % tag), "w"), indent=1)
    print("profiles/ updated from", EV)


if __name__ == "__main__":
    main()
