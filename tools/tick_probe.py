import sys, json
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import bench_legs
from ergodic_exploration_amd import capi
r = bench_legs.tick_legs(torch, capi, np)
for c in r["tick_kernels"]["cases"]:
    print({k: (round(v, 2) if isinstance(v, float) else v) for k, v in c.items() if not k.endswith("_step") and not k.endswith("_pose")})
ft = r["fleet_tick"]
print({k: (round(v, 2) if isinstance(v, float) else v) for k, v in ft.items() if k not in ("note", "bounds")})
