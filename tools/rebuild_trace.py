#!/usr/bin/env python3
"""50 enqueue-only configTarget rebuilds per BASELINE grid (Gaussian target), for rocprofv3 --kernel-trace --stats:
   rocprofv3 --kernel-trace --stats -d gpurun_out/rebuild_trace -o t -- python3 tools/rebuild_trace.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ergodic_exploration_amd import capi  # noqa: E402

MEANS, SIGMAS = [[2.5, 2.5], [8.5, 2.5]], [[1.5, 1.5], [1.5, 1.5]]
impl = int(sys.argv[1]) if len(sys.argv) > 1 else 0
capi.set_option(capi.OPT_REBUILD_IMPL, impl)
st = torch.cuda.Stream()
for Kc, lxc, lyc in ((10, 12.0, 6.0), (20, 25.5, 25.5), (30, 102.3, 102.3)):
    e2 = capi.Engine(capi.make_config(capi.MODEL_OMNI, 0.1, 2.0, 0.1, 1.0, Kc, np.eye(3), [-1] * 3, [1] * 3))
    e2.set_target_gaussians(MEANS, SIGMAS)
    for i in range(50):
        e2.config_domain_async((0.0, lxc + 0.1 * (i % 2), 0.0, lyc), stream=st.cuda_stream)
    st.synchronize()
    e2.close()
