"""The SURVEY.md 8(f) kernels and the fleet tick by themselves (bench.py's tick_legs), for rocprofv3:

    rocprofv3 --kernel-trace --stats -d gpurun_out/tick_prof -o tick -- python3 tools/tick_point.py
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ergodic_exploration_amd import capi  # noqa: E402

if __name__ == "__main__":
    torch.cuda.set_device(0)
    print(json.dumps(bench.tick_legs(torch, capi, np), indent=1))
