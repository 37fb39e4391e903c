P='import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("%-50s %6d L=%2d %8.3f us/pass %7.3f us/4096 frac %.4f" % (r["config"], r["agents"], r["lanes_per_agent"], 1e3*r["ms_per_pass"], r["us_per_4096_agents"], r["roofline"]["frac"]))'
for rep in 1 2 3; do
for v in "" ${AB_VARIANT:-_nopad}; do
  echo "== variant [$v] rep $rep"
  for c in "explore_omni.yaml as shipped, chip-filling batch" "configs[1], chip-filling batch" "configs[0], chip-filling batch"; do
    EEA_LIB_VARIANT=$v python3 tools/other_config_point.py --case "$c" 2>/dev/null | python3 -c "$P"
  done
done
done
