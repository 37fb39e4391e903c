"""The ONE line `bench.py` leaves as the last line of stdout, and the detail file beside it.

The driver keeps a bounded tail of stdout and parses its last line: round 5's line had grown to 22.5 KB, its head fell
off the tail and the round went unmeasured.  So the line is built HERE from the full result dictionary by selection,
never by accretion: a fixed set of fields, numbers rounded to 6 significant digits, no prose; `compact()` asserts the
bound (`MAX_LINE_BYTES`) so that a leg added to `bench.py` cannot make the line grow -- everything else goes to
`bench_detail.json` (`write_detail`).  `tests/test_bench_line.py` checks the bound on the largest dictionary a run has
produced and on the dry-run path.
"""
import json
import math
import os
import re

MAX_LINE_BYTES = 4096          # "final line <= 4 KB"
DETAIL_NAME = "bench_detail.json"


def _r(x, sig=6):
    """numbers to `sig` significant digits (ints and non-numbers untouched)"""
    if isinstance(x, bool) or not isinstance(x, float):
        return x
    if x == 0.0 or not math.isfinite(x):
        return x if math.isfinite(x) else None
    return float("%.*g" % (sig, x))


def _pick(d, keys, sig=6):
    if not isinstance(d, dict):
        return None
    return {k: _r(d[k], sig) for k in keys if k in d and d[k] is not None}


SHORT_COLS = ("config", "agents", "lanes_per_agent", "us_per_4096", "frac", "kernel_avg_us_profiled", "frac_profiled", "variant")


def _short_horizons(other):
    """one row per BASELINE shape and batch size of bench.py's other_configs, as {"cols": SHORT_COLS, "rows": [[...]]}: config,
    agents, lanes per agent (64 = one wavefront per agent), us per 4096 agents, fraction of the dtype's vector peak, the
    rocprofv3 kernel average of the same launch form and the fraction it gives when profiles/ holds one (else null), and
    what makes the row a variant of its config (dtype / model / replay memory)"""
    rows = []
    for c in (other or {}).get("cases", []) if isinstance(other, dict) else []:
        if not isinstance(c, dict):
            continue
        name = str(c.get("config", ""))
        m = re.match(r"configs\[\d\]", name)
        variant = []
        if c.get("dtype") not in (None, "f64"):
            variant.append(c["dtype"])
        if c.get("kinematics") == "omni" and m and m.group(0) == "configs[3]":
            variant.append("omni")
        if c.get("n_mem"):
            variant.append("n_mem=%d" % c["n_mem"])
        rows.append([m.group(0) if m else ("yaml K10 T50" if "yaml" in name else name[:16]), c.get("agents"), c.get("lanes_per_agent"),
                     _r(c.get("us_per_4096_agents"), 4), _r((c.get("roofline") or {}).get("frac"), 3),
                     _r(c.get("kernel_avg_us_profiled"), 4), _r(c.get("frac_profiled"), 3), " ".join(variant) or None])
    return {"cols": list(SHORT_COLS), "rows": rows} if rows else None


def _exchange(ex):
    if not isinstance(ex, dict):
        return None
    if "error" in ex and "consensus_allreduce" not in ex:
        return {"error": str(ex["error"])[:200]}
    o = {"backend": str(ex.get("backend", ""))[:60]}
    for k in ("rccl_nranks", "world"):
        if ex.get(k) is not None:
            o[k] = ex[k] if not isinstance(ex[k], float) else _r(ex[k])
    ca = ex.get("consensus_allreduce")
    if isinstance(ca, dict):
        o.update({"lag": ca.get("lag_passes"), "pass_ms": _r(ca.get("pass_ms")),
                  "vs_headline": _r(ca.get("pass_ms_vs_headline"), 4),
                  "vs_single_launch_pass": _r(ca.get("pass_ms_vs_single_launch_pass"), 4),
                  "protocol": str(ca.get("protocol", "")).split(" (")[0][:32],
                  "timeouts": sum(int(v.get("agents_timed_out", 0)) for v in (ca.get("by_lag") or {}).values()
                                  if isinstance(v, dict))})
    ag = ex.get("allgather_ck")
    if isinstance(ag, dict):
        o["allgather_pass_ms"] = _r(ag.get("pass_ms"))
    cpp = ex.get("cpp_host_loop")
    if isinstance(cpp, dict):
        rows = []
        packed = []
        for c in cpp.get("cases", []):
            if not isinstance(c, dict):
                continue
            if (c.get("lanes_per_agent") or 64) < 64:   # the short-horizon cases have their own plain pass: their own block
                packed.append([c.get("agents"), c.get("horizon_steps"), c.get("records_per_pass"), _r(c.get("plain_us_per_pass"), 4),
                               _r(c.get("consensus_us_per_pass"), 4), _r(c.get("ratio"), 4)])
                continue
            form = str(c.get("form") or c.get("consuming_groups", ""))
            form = "plan" if "eea_consensus_plan" in form else ("stream-ordered" if form.startswith("all stream") else
                                                                  ("device-bound" if form.startswith("all device") else
                                                                   ("gated" if form.startswith("all gated") else "hybrid")))
            rows.append({"form": form, "collective_kernel": c.get("collective_kernel_in_exchange"), "lag": c.get("lag"),
                         "plain_us": _r(c.get("plain_us_per_pass"), 4), "consensus_us": _r(c.get("consensus_us_per_pass"), 4),
                         "ratio": _r(c.get("ratio"), 4), "host_us": _r(c.get("host_us_per_pass_consensus"), 4),
                         "timeouts": c.get("agents_timed_out")})
        # (the event-ordered per-call form and the hybrid of round 5 stay in bench_detail.json: superseded by the gated form)
        keep = [r for r in rows if r["form"] in ("device-bound", "gated", "plan")][:4]
        cols = ("form", "collective_kernel", "lag", "consensus_us", "ratio", "host_us", "timeouts")
        if keep:
            o["cpp_host_loop"] = {"plain_us": keep[0]["plain_us"], "cols": list(cols), "rows": [[r[k] for k in cols] for r in keep]}
        if packed:
            o["packed_gated"] = {"cols": ["agents", "T", "records", "plain_us", "consensus_us", "ratio"], "rows": packed[:2]}
    return o


def compact(out):
    """the driver's line from the full result dictionary `out` (see module docstring); raises if it would exceed the bound"""
    line = {k: _r(out[k]) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                   "scaling", "vs_baseline", "dtype", "data") if k in out}
    cfg = out.get("config")
    if isinstance(cfg, dict):
        line["config"] = {"workload": str(cfg.get("workload", "")).split(";")[0][:140],
                          **(_pick(cfg, ("agents_per_gpu", "num_basis", "horizon_steps", "kinematics", "passes_per_step",
                                         "agent_groups", "steps_per_launch", "parallelism")) or {})}
    for k in ("timed_region_s", "ms_per_pass"):
        if k in out:
            line[k] = _r(out[k])
    pr = out.get("per_rank")
    if isinstance(pr, dict) and isinstance(pr.get("values"), list) and len(pr["values"]) > 1:
        line["per_rank"] = {"timed_region_s": [_r(float(x), 5) for x in (pr.get("timed_region_s") or [])][:8],
                            "values": [_r(float(x), 5) for x in pr["values"]][:8]}
    rf = out.get("roofline")
    if isinstance(rf, dict):
        line["roofline"] = _pick(rf, ("bound", "achieved", "peak", "unit", "frac", "traffic", "launch_ms", "flops_per_launch", "passes_per_launch",
                                      "agents_per_launch", "concurrent_launches", "kernel_avg_us_profiled", "frac_profiled",
                                      "issue_bound_us", "frac_of_issue_bound", "mfma_busy_frac",
                                      "wait_inst_any_over_wave_cycles"))
        line["roofline"]["kernel"] = str(rf.get("kernel", ""))[:24].split(" (")[0]
        if "traffic" not in line["roofline"]:
            line["roofline"]["traffic"] = None
    rh = out.get("roofline_hbm")
    if isinstance(rh, dict):
        line["roofline_hbm"] = _pick(rh, ("algorithmic_rate", "counter_rate", "peak", "unit", "algorithmic_frac"))
    sl = out.get("single_launch_per_pass")
    if isinstance(sl, dict):
        line["single_launch_per_pass"] = _pick(sl, ("ms_per_pass", "frac"))
    rp = out.get("roofline_phik")
    if isinstance(rp, dict):
        line["roofline_phik"] = _pick(rp, ("bound", "achieved", "peak", "unit", "frac", "traffic", "launch_ms", "error"))
    cb = out.get("cpu_baseline")
    if isinstance(cb, dict):
        line["cpu_baseline"] = _pick(cb, ("value", "unit", "cores", "kind", "cpu_model", "nproc", "cflags", "error"))
        line["cpu_baseline"]["sample"] = str(cb.get("sample", ""))[:100]
    ca = out.get("cpu_baseline_all_cores")
    if isinstance(ca, dict):
        line["cpu_baseline_all_cores"] = _pick(ca, ("value", "cores"))
    sh = _short_horizons(out.get("other_configs"))
    if sh:
        line["short_horizons"] = sh
    ex = _exchange(out.get("exchange"))
    if ex:
        line["exchange"] = ex
    ft = out.get("fleet_tick")
    if isinstance(ft, dict):
        line["fleet_tick"] = _pick(ft, ("robots", "us_per_tick", "us_per_tick_unchanged_grid", "us_per_tick_moving_robots", "error"), 5)
    lm = out.get("latency_mode")
    if isinstance(lm, dict):
        line["latency_mode"] = _pick(lm, ("value", "us_per_call"), 5)
    srt = out.get("single_robot_tick")
    if isinstance(srt, dict) and isinstance(srt.get("cpp_host"), dict):
        line["single_robot_us"] = [[c.get("horizon_steps"), _r(c.get("launch_us_per_call"), 4), _r(c.get("resident_us_per_call"), 4)]
                                   for c in srt["cpp_host"].get("cases", []) if isinstance(c, dict)][:6]
    gt = out.get("grid_tile")
    if isinstance(gt, dict):
        line["grid_tile"] = _pick(gt, ("rows_per_rank", "us_per_rebuild_back_to_back", "ok", "max_abs_err_vs_untiled", "error"), 5)
    for k in ("dryrun", "gpus_arg"):
        if k in out:
            line[k] = out[k]
    line["detail"] = DETAIL_NAME
    text = json.dumps(line, separators=(",", ":"))
    # by construction, not by luck: drop the optional blocks, least important first, until the line fits
    for k in ("single_robot_us", "grid_tile", "latency_mode", "fleet_tick", "roofline_hbm", "cpu_baseline_all_cores"):
        if len(text) <= MAX_LINE_BYTES:
            break
        line.pop(k, None)
        text = json.dumps(line, separators=(",", ":"))
    while len(text) > MAX_LINE_BYTES and line.get("short_horizons", {}).get("rows"):
        line["short_horizons"]["rows"].pop()
        text = json.dumps(line, separators=(",", ":"))
    if len(text) > MAX_LINE_BYTES:
        raise AssertionError("bench line is %d bytes (> %d)" % (len(text), MAX_LINE_BYTES))
    return text


def write_detail(out, root):
    """the full dictionary (every leg, every note) next to bench.py, and under gpurun_out/ when that directory exists (it is
    what travels back from a GPU box); never fails the run"""
    paths = [os.path.join(root, DETAIL_NAME)]
    scratch = os.path.join(root, "gpurun_out")
    if os.path.isdir(scratch):
        paths.append(os.path.join(scratch, DETAIL_NAME))
    for p in paths:
        try:
            with open(p, "w") as f:
                json.dump(out, f, indent=1)
                f.write("\n")
        except OSError:
            pass
