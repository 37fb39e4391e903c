"""ONE leg of bench.py's `other_configs` as a stand-alone program, in exactly that leg's launch form (two agent groups on two
streams x `--steps-per-launch` receding-horizon steps per launch) behind a clock spin-up -- the program tools/r06_pack_profile.sh
puts after `rocprofv3 --kernel-trace --stats --`.  Prints one JSON line: the leg's record incl. `launches_timed` (the number of
control dispatches of the timed region = the LAST dispatches of the trace).

    python3 tools/other_config_point.py --case "configs[1], chip-filling batch"
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", required=True, help="the leg's name in bench.py's other_config_legs")
    ap.add_argument("--steps-per-launch", type=int, default=50)
    ap.add_argument("--agents", type=int, default=4096, help="batch of the cases that do not name their own")
    ap.add_argument("--spinup-s", type=float, default=1.0)
    ap.add_argument("--lanes", type=int, default=0, help="force EEA_OPT_AGENT_LANES (0: the engine's cost model)")
    a = ap.parse_args()
    import numpy as np
    import torch
    import bench_legs as bench
    from ergodic_exploration_amd import capi
    if a.lanes:
        capi.set_option(capi.OPT_AGENT_LANES, a.lanes)
    rec = bench.other_config_legs(a, torch, capi, np, a.steps_per_launch, only=[a.case], spinup_s=a.spinup_s)["cases"]
    assert len(rec) == 1, "no such case: %r" % a.case
    print(json.dumps(rec[0]))


if __name__ == "__main__":
    main()
