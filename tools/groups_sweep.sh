for v in "" "_w5"; do for g in 1 2 4 8; do
out=$(EEA_LIB_VARIANT=$v python3 bench.py --steps 100 --warmup 10 --cpu-seconds 0 --agent-groups $g 2>/dev/null | tail -1)
echo "variant '$v' groups $g $(echo "$out" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.4g opt/s  step %.2f us  launch %.2f us enqueue %.1f us" % (d["value"], 1e3*d["ms_per_step"], 1e3*d["roofline"]["launch_ms"], d["host_enqueue_us_per_step"]))')"
done; done
