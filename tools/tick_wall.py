#!/usr/bin/env python3
"""Wall time per Exploration tick of the non-ROS entry points (host mirror on the C ABI), as the entry
point times its own loop (1000 ticks; the first tick includes the phi_k build)."""
import os
import subprocess
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B = os.path.join(ROOT, "ergodic_exploration_amd", "host", "build")
C = os.path.join(ROOT, "ergodic_exploration_amd", "host", "config")


def run(name, ticks, extra):
    out = subprocess.run([os.path.join(B, "exploration_" + name), "--params", os.path.join(C, "explore_%s.yaml" % name),
                          "--ticks", str(ticks)] + extra, capture_output=True, text=True, check=True).stdout
    src = [l.split()[-1] for l in out.splitlines() if l.startswith("tick")]
    loop = [l for l in out.splitlines() if l.startswith("# loop")][0]
    import re
    nums = [float(v) for v in re.findall(r"([0-9.]+) us", loop)]
    return nums[0], nums[1], {s: src.count(s) for s in set(src)}


for name in ("omni", "cart"):
    for label, extra in (("free map", []), ("obstacle", ["--obstacle", "2.4", "0.2", "3.0", "2.6"])):
        us, total, mix = run(name, 1000, extra)
        print("exploration_%s %-9s %7.1f us per tick() (%6.1f with printing)   %s" % (name, label, us, total, mix))
