"""The driver parses the LAST stdout line of bench.py from a bounded tail (round 5's 22.5 KB line lost its head there and
the round went unmeasured): the line is a fixed selection of fields, <= 4 KB by construction.  CPU tests, no GPU."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench_line  # noqa: E402

RECORDED = [os.path.join(ROOT, "profiles", n) for n in ("r05_bench.json", "r06_bench_detail.json")]
RECORDED = [p for p in RECORDED if os.path.exists(p)]

REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline")


@pytest.mark.parametrize("path", RECORDED, ids=[os.path.basename(p) for p in RECORDED])
def test_compact_line_of_a_recorded_run_is_small_and_complete(path):
    with open(path) as f:
        full = json.load(f)
    text = bench_line.compact(full)
    assert "\n" not in text and len(text) <= bench_line.MAX_LINE_BYTES < 6000
    rec = json.loads(text)
    for k in REQUIRED:
        assert k in rec, k
    assert rec["value"] == pytest.approx(full["value"], rel=1e-5)
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(rec["roofline"])
    assert rec["roofline"]["frac"] == pytest.approx(full["roofline"]["achieved"] / full["roofline"]["peak"], rel=1e-4)
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(rec["cpu_baseline"])
    assert "workload" in rec["config"] and "model" not in rec["config"]
    if "r06" in os.path.basename(path):
        # round 6: the consensus pass of the packed kernel from the C++ host loop -- one sum record per wavefront first, per agent second
        pg = rec["exchange"]["packed_gated"]
        rows = [dict(zip(pg["cols"], r)) for r in pg["rows"]]
        assert rows[0]["records"] < rows[0]["agents"] == rows[1]["records"] and rows[0]["ratio"] < rows[1]["ratio"]
    sh = rec["short_horizons"]
    assert sh["cols"][:5] == ["config", "agents", "lanes_per_agent", "us_per_4096", "frac"] and len(sh["rows"]) >= 6
    assert all(len(r) == len(sh["cols"]) for r in sh["rows"])
    assert "note" not in text   # no prose in the line


def test_compact_line_cannot_grow_with_new_legs():
    """whatever a later round adds to the result dictionary, the line stays a selection: unknown keys and long notes do
    not reach it, and when the selected blocks themselves are too many the optional ones go first"""
    with open(RECORDED[0]) as f:
        full = json.load(f)
    full["a_new_leg"] = {"note": "x" * 20000, "cases": [{"k": i, "note": "y" * 500} for i in range(50)]}
    full["other_configs"]["cases"] = full["other_configs"]["cases"] * 6
    full["exchange"]["cpp_host_loop"]["cases"] = full["exchange"]["cpp_host_loop"]["cases"] * 10
    text = bench_line.compact(full)
    assert len(text) <= bench_line.MAX_LINE_BYTES
    rec = json.loads(text)
    for k in REQUIRED:
        assert k in rec, k


def test_bench_emits_the_compact_line_and_the_detail_file(tmp_path):
    """bench.py's own emission path (dry run: no GPU) fed a recorded full dictionary: stdout's last line is the compact
    line, the full record lands in bench_detail.json"""
    env = dict(os.environ, EEA_BENCH_DRYRUN="1", EEA_BENCH_DRYRUN_FROM=RECORDED[0], EEA_BENCH_DETAIL_DIR=str(tmp_path))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-grid-tile"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    last = r.stdout.rstrip("\n").splitlines()[-1]
    assert len(last) < 6000 and len(last) <= bench_line.MAX_LINE_BYTES
    rec = json.loads(last)
    assert rec["dryrun"] is True and "roofline" in rec and "cpu_baseline" in rec and rec["detail"] == bench_line.DETAIL_NAME
    with open(tmp_path / bench_line.DETAIL_NAME) as f:
        detail = json.load(f)
    assert "other_configs" in detail and len(json.dumps(detail)) > 10000
