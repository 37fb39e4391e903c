#!/usr/bin/env python3
"""Prints the worst |GPU - oracle| per stage for the parity configurations (run on the GPU box);
the output is committed under profiles/ as the measured parity margin."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

os.environ["EEA_PRINT_WORST"] = "1"   # the full-size tests print their measured worst differences

from ergodic_exploration_amd import capi  # noqa: E402
from tests import test_gpu_control_parity as t  # noqa: E402
from tests import test_gpu_round2_parity as t2  # noqa: E402

CASES = [("omni", 5, 0.5, 0.1, 0), ("simple_cart", 10, 2.0, 0.1, 0), ("omni", 10, 5.0, 0.1, 7),
         ("simple_cart", 10, 5.0, 0.1, 100), ("omni", 10, 20.0, 0.1, 0), ("simple_cart", 10, 20.0, 0.1, 100),
         ("omni", 7, 30.0, 0.1, 5), ("omni", 30, 50.0, 0.1, 0)]
print("fp64: worst abs difference kernel vs oracle over B agents x calls (per case: the run's own line with the stage\n"
      "magnitudes, then the summary)")
for model, K, hor, dt, n_mem in CASES:
    B = 1 if K >= 30 else 4
    w = t.run_batch_vs_oracle(model, K, hor, dt, B=B, n_mem=n_mem, calls=2, seed=21)
    print("%-12s K=%-2d T=%-3d n_mem=%-3d " % (model, K, int(abs(hor / dt) + 1e-9), n_mem) +
          " ".join("%s=%.1e" % (k, v) for k, v in sorted(w.items())))
bounds = (0.0, 25.5, 0.0, 25.5)
means, sigmas = [[6.0, 6.0], [19.0, 12.0]], [[3.0, 3.0], [3.0, 3.0]]
for prec, name, tol, tck in ((capi.PREC_F64, "f64", 1e-9, 1e-11), (capi.PREC_F32, "f32", 5e-4, 1e-5)):
    w = t.run_batch_vs_oracle("omni", 20, 5.0, 0.02, B=2, n_mem=0, calls=2, seed=3, bounds=bounds, means=means,
                              sigmas=sigmas, precision=prec, tol=tol, tol_ck=tck)
    print("config3 %s  K=20 T=250 " % name + " ".join("%s=%.1e" % (k, v) for k, v in sorted(w.items())))
print("BASELINE config 4 at full size (4096 agents, K=10, T=200): (|u0 diff|, |ut diff|) after call 1 and call 2")
for model in ("simple_cart", "omni"):
    t2.test_config4_full_size_against_oracle(model)
t2.test_config4_full_size_f32_against_f64_oracle()
print("BASELINE config 5 end to end (1024^2 occupancy -> phi_k K=30 -> control T=500)")
t2.test_config5_end_to_end_against_oracle()
print("round 4: the TIMED instances (no stage outputs: STAGES = false) against the oracle, c_k / ut / u0")
from tests import test_gpu_timed_instances as tt  # noqa: E402
for model, K, hor, dt, n_mem in CASES[:6] + [("omni", 10, 19.3, 0.1, 40), ("simple_cart", 10, 19.9, 0.1, 0)]:
    t.run_batch_vs_oracle(model, K, hor, dt, B=4, n_mem=n_mem, calls=2, seed=21, stages=False)
for prec, name, tol, tck in ((capi.PREC_F64, "f64", 1e-9, 1e-11), (capi.PREC_F32, "f32", 5e-4, 1e-5)):
    t.run_batch_vs_oracle("omni", 20, 5.0, 0.02, B=2, n_mem=0, calls=2, seed=3, bounds=bounds, means=means,
                          sigmas=sigmas, precision=prec, tol=tol, tol_ck=tck, stages=False)
print("round 4: the consensus leg's exact form (no stages, d_ck_rec out, one sum record in) against the oracle's shared-c_k switch")
for args in (("simple_cart", 10, 20.0, 0, 1), ("omni", 10, 19.7, 40, 2), ("omni", 5, 19.3, 0, 1), ("omni", 20, 5.0, 0, 1),
             ("omni", 30, 6.0, 0, 1)):
    tt.test_consensus_leg_exact_form_against_oracle(*args, capi.PREC_F64)
print("... and on the fp32 engine (bars 5e-4 / 1e-5): K = 20 and K = 10 with replay memory, T = 50 (at T = 200 the fp32 co-state error, relative to |rho| ~ 1e5, is what the controls inherit: the stage-wise tests use the rho-relative bar there)")
for args in (("omni", 20, 5.0, 33, 1, capi.PREC_F32), ("simple_cart", 10, 5.0, 7, 1, capi.PREC_F32)):
    tt.test_consensus_leg_exact_form_against_oracle(*args)
print("round 4: fp32 K = 20 (outer-product contraction, gradient packed over pairs of steps), 1 .. 4 steps per lane")
for steps, n_mem in ((250, 0), (200, 100), (129, 65), (64, 33), (37, 1)):
    for stages in (True, False):
        t.run_batch_vs_oracle("omni", 20, steps * 0.02, 0.02, B=3, n_mem=n_mem, calls=2, seed=2000 + steps,
                              precision=capi.PREC_F32, tol=5e-4, tol_ck=1e-5, stages=stages)
for dt in (1.0, 2.0):
    print("large step increments dt=%g (bars relative to max(1, |stage|))" % dt)
    t.test_small_and_large_step_increments(dt)
