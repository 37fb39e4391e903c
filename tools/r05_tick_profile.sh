set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/tick_prof
python3 tools/tick_point.py > gpurun_out/tick_prof/tick_legs.json 2> gpurun_out/tick_prof/tick_legs.err; tail -3 gpurun_out/tick_prof/tick_legs.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tick_prof/trace -o tick -- python3 tools/tick_point.py > gpurun_out/tick_prof/trace.log 2>&1
python3 - <<'PY'
import csv, glob, json
d = json.load(open('gpurun_out/tick_prof/tick_legs.json'))
out = ["# bench.py tick_legs (device time from HIP events; un-profiled run)"]
for c in d["tick_kernels"]["cases"]:
    out.append("%-13s P=%-6d collision_check %8.2f us  validate_control %8.2f us  dwa vref %9.2f us  dwa traj %9.2f us" %
               (c["implementation"], c["poses"], c["collision_check_us"], c["validate_control_us"], c["dwa_vref_us"], c["dwa_traj_us"]))
f = d["fleet_tick"]
out.append("fleet tick, %d robots (K=10, T=%d): %.2f us per tick, %.2f us on an unchanged grid (grid_epoch); control_batch alone %.2f us; sources of the last tick %s" %
           (f["robots"], f["horizon_steps"], f["us_per_tick"], f["us_per_tick_unchanged_grid"], f["control_batch_alone_us"], f["sources_last_tick"]))
out.append("")
out.append("# rocprofv3 --kernel-trace --stats -- python3 tools/tick_point.py  (kernel stats of the same program)")
for p in glob.glob('gpurun_out/tick_prof/trace/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(p)):
        out.append("%-100s calls=%-6s avg_ns=%-10s total_ns=%-12s pct=%s" % (r["Name"][:100], r["Calls"], r["AverageNs"], r["TotalDurationNs"], r["Percentage"]))
open('gpurun_out/tick_prof/r05_tick_kernels.txt', 'w').write("\n".join(out) + "\n")
print("\n".join(out))
PY
