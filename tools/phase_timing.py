#!/usr/bin/env python3
"""Per-phase latency of the control kernel (K = 10, T = 200, fp64) from the instrumented A/B build
(make -C ergodic_exploration_amd/csrc AB=1): mean shader-clock cycles every wavefront spends between the
phase stamps.  Default: the wavefront-per-agent kernel ([agent][16] stamps); EEA_PHASE_KERNEL=workgroup: the
workgroup-per-agent kernel ([agent][4 waves][16]) through eea_set_option(EEA_OPT_CONTROL_KERNEL, 1)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("EEA_LIB_VARIANT", "_ab")
from ergodic_exploration_amd import capi  # noqa: E402

PHASES_WG = ["load+shift controls", "heading scan", "heading sincos", "position scan", "basis sincos",
             "tables + MFMA c_k", "reduce c_k, D", "gradient", "rho01 scan", "rho2 scan", "update+store"]
PHASES_WAVE = ["load+shift controls", "heading scan", "heading sincos + position scan", "basis sincos + barrier",
               "tables + MFMA c_k", "D = lambda (c - phi)", "gradient", "rho scans", "update + store"]


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    force_wg = os.environ.get("EEA_PHASE_KERNEL") == "workgroup"
    if force_wg:
        capi.set_option(capi.OPT_CONTROL_KERNEL, 1)
    wg = force_wg or int(os.environ.get("EEA_PHASE_K", "10")) > 20 or \
        float(os.environ.get("EEA_PHASE_HORIZON", "20.0")) > 25.6
    model = capi.MODEL_SIMPLE_CART
    lim = np.array([1.0, 0.0, 2.0])
    K = int(os.environ.get("EEA_PHASE_K", "10"))
    horizon = float(os.environ.get("EEA_PHASE_HORIZON", "20.0"))  # dt = 0.1
    f32 = os.environ.get("EEA_PHASE_PRECISION", "f64") == "f32"
    eng = capi.Engine(capi.make_config(model, 0.1, horizon, 0.1, 1.0, K, np.diag([1.0, 0.0, 2.0]), -lim, lim,
                                       precision=capi.PREC_F32 if f32 else capi.PREC_F64))
    eng.set_target_gaussians([[2.5, 2.5], [8.5, 2.5]], [[1.5, 1.5], [1.5, 1.5]])
    eng.config_domain((-1.0, 11.0, -1.0, 5.0))
    T, K2 = eng.T, eng.K2
    rng = np.random.default_rng(1)
    poses = np.stack([rng.uniform(-0.5, 10.5, B), rng.uniform(-0.5, 4.5, B), rng.uniform(-3, 3, B)], 1)
    tdt = torch.float32 if f32 else torch.float64
    d_pose = torch.as_tensor(poses).to(tdt).cuda()
    d_ut = torch.zeros((B, T, 3), dtype=tdt, device="cuda")
    d_u0 = torch.empty((B, 3), dtype=tdt, device="cuda")
    for _ in range(int(os.environ.get("EEA_PHASE_WARMUP", "5"))):  # a long warm-up = the clock under sustained load
        eng.control_batch(B, d_pose, d_ut, d_u0)
    stamps = torch.zeros((B, 4, 16), dtype=torch.int64, device="cuda")
    eng.debug_phase_timing(B, d_pose, d_ut, d_u0, stamps)
    torch.cuda.synchronize()
    if wg:
        s = stamps.cpu().numpy()[:, :, :12].astype(np.float64)
        names = PHASES_WG
    else:
        s = stamps.cpu().numpy().reshape(-1)[:B * 16].reshape(B, 1, 16)[:, :, :10].astype(np.float64)
        names = PHASES_WAVE
    if not wg:
        raw = stamps.cpu().numpy().reshape(-1)[:B * 16].reshape(B, 16)
        rt = raw[:, 10:12].astype(np.float64) * 10.0  # 100 MHz counter -> ns
        life_ns = rt[:, 1] - rt[:, 0]
        cyc = (raw[:, 9] - raw[:, 0]).astype(np.float64)
        print("shader clock while the wavefronts ran: %.2f GHz (mean of cycles / lifetime); launch span %.2f us, "
              "wave lifetime %.2f us mean" % ((cyc / life_ns).mean(), (rt[:, 1].max() - rt[:, 0].min()) * 1e-3,
                                              life_ns.mean() * 1e-3))
        hw = raw[:, 12]
        key = (hw >> 32 << 16) | (((hw >> 13) & 7) << 12) | (((hw >> 12) & 1) << 11) | (((hw >> 8) & 15) << 4) | ((hw >> 4) & 3)
        order = {}
        for a in range(B):
            order.setdefault(int(key[a]), []).append(a)
        per = np.array([len(v) for v in order.values()])
        print("SIMDs used %d; wavefronts per SIMD min %d mean %.2f max %d" % (len(order), per.min(), per.mean(), per.max()))
        # finishing order within a SIMD (shader clock of one XCD: comparable within the SIMD)
        rows = []
        both = {"mfma": 0.0, "grad": 0.0, "span": 0.0}
        for v in order.values():
            if len(v) != 4:
                continue
            st = raw[v][:, :10].astype(np.float64)
            t0 = st[:, 0].min()
            fin = np.sort(st[:, 9] - t0)
            rows.append(fin)
            # time with >= 2 wavefronts of the SIMD inside the same exclusive phase
            for name, (a, b_) in (("mfma", (4, 5)), ("grad", (6, 7))):
                ev = sorted([(x, 1) for x in st[:, a]] + [(x, -1) for x in st[:, b_]])
                n, last, acc = 0, 0.0, 0.0
                for x, dn in ev:
                    if n >= 2:
                        acc += x - last
                    n += dn
                    last = x
                both[name] += acc
            both["span"] += st[:, 9].max() - t0
        if rows:
            rows = np.array(rows)
            print("SIMDs with 4 wavefronts: %d; finish times after the SIMD's first start (cycles): %s" %
                  (len(rows), " / ".join("%.0f" % x for x in rows.mean(0))))
            print("  share of the SIMD's span with >= 2 wavefronts inside the contraction: %.1f %%, inside the gradient: "
                  "%.1f %%" % (100 * both["mfma"] / both["span"], 100 * both["grad"] / both["span"]))
    d = np.diff(s, axis=2)
    total = s[:, :, -1] - s[:, :, 0]
    print("agents %d: wave lifetime mean %.0f cycles (min %.0f max %.0f); launch span %.0f cycles; first start "
          "spread %.0f cycles" % (B, total.mean(), total.min(), total.max(), s[:, :, -1].max() - s[:, :, 0].min(),
                                  s[:, :, 0].max() - s[:, :, 0].min()))
    print("%-34s %10s %7s" % ("phase", "cycles", "share"))
    m = d.mean((0, 1))
    for name, v in zip(names, m):
        print("%-34s %10.0f %6.1f%%" % (name, v, 100 * v / m.sum()))
    eng.close()


if __name__ == "__main__":
    main()
