#!/usr/bin/env python3
"""Static instruction budget of the wavefront-per-agent control kernel at the metric point (fp64, SimpleCart, K = 10,
T = 200, no replay memory): per phase, how many instructions of which class one wavefront (= one agent) issues.

Compiles tools/ab/budget_kernel.hip -- the product's kernel text (csrc/control_wave_impl.hpp, copied with the shape fixed
at compile time by the substitutions below) with the A/B library's phase stamps switched on -- to gfx950 assembly and
walks the executed path:
  * exec-masked skips (s_cbranch_execz) fall through, out-of-line bodies (s_cbranch_execnz) are entered, except the
    huge-argument path of sin/cos (v_fract: |t| >= 2^50, never taken by headings or in-domain positions);
  * the few wavefront-uniform data-dependent branches left (small-increment fast paths, rejected-twist exit) are tried
    both ways; the path that passes every phase stamp with the fewest vector instructions is the one the metric
    workload runs (both fast paths on, no rejected agent).
Phases are delimited by the stamps' s_memtime instructions.  The totals are checked against the hardware counters of
the real kernel (SQ_INSTS_VALU etc., profiles/r03_*_summary.txt) by the caller.
Usage: python tools/isa_budget.py [--keep /tmp/budget.s]
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PHASES = ["load + shift controls", "heading increments + scan", "heading sin/cos, Simpson increments, position scan",
          "basis sin/cos + barrier gradient", "tables + contraction (c_k)", "D = lambda (c - phi)", "gradient (4 steps)",
          "co-state scans", "update + store"]


def classify(ins):
    op = ins.split()[0]
    if op.startswith("v_mfma"):
        return "mfma"
    if op in ("v_fma_f64", "v_fmac_f64_e32", "v_fmac_f64_e64", "v_fmac_f64"):
        return "v_fma64"
    if op.startswith("v_mul_f64") or op.startswith("v_add_f64"):
        return "v_muladd64"
    if op.startswith("v_"):
        return "v_other"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_"):
        return "vmem"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"):
        return "wait"
    if op.startswith("s_load") or op.startswith("s_memtime") or op.startswith("s_memrealtime"):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    keep = sys.argv[sys.argv.index("--keep") + 1] if "--keep" in sys.argv else None
    tmp = tempfile.mkdtemp()
    out = keep or os.path.join(tmp, "budget.s")
    csrc = os.path.join(ROOT, "ergodic_exploration_amd", "csrc")
    # the kernel text with the metric point's shape as compile-time constants
    text = open(os.path.join(csrc, "control_wave_impl.hpp")).read()
    subs = [("const ControlParams<R> p, const unsigned B, const int S, const int rollout_only)\n{\n",
             "const ControlParams<R> p_in, const unsigned B, const int S_in, const int rollout_in)\n{\n"
             "  ControlParams<R> p = p_in;\n  p.mem_cols = nullptr;\n  p.n_mem = nullptr;\n  p.mem_stride = 0;\n"
             "  p.traj = p.edx = p.bdx = p.rhot = nullptr;\n  p.ck = nullptr;\n  p.ck_rec = nullptr;\n"
             "  p.ck_shared = nullptr;\n  p.ck_shared_parts = 0;\n  p.done = nullptr;\n  p.K = 10;\n"
             "  constexpr int S = 4;\n  constexpr int rollout_only = 0;\n"),
            ("  const int T = p.T;\n", "  constexpr int T = 200;\n")]
    for a, b in subs:
        assert text.count(a) == 1, a
        text = text.replace(a, b)
    with open(os.path.join(tmp, "budget_impl.hpp"), "w") as f:
        f.write(text)
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-function", "-fno-slp-vectorize",
           "-I", tmp, "-I", csrc, "-include", os.path.join(ROOT, "tools", "ab", "wave_stamps.hpp"), "-S", "--cuda-device-only",
           "-o", out, os.path.join(ROOT, "tools", "ab", "budget_kernel.hip")]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    src = open(out).read().split("\n")
    start = next(i for i, l in enumerate(src) if re.match(r"^_ZN3eea4wave24control_wave_kernel_leanIdLi1ELi10ELb0ELi4E.*:", l))
    end = next(i for i, l in enumerate(src) if l.startswith(".Lfunc_end") and i > start)
    blocks, order, cur = {}, [], "entry"
    blocks[cur] = []
    order.append(cur)
    for l in src[start + 1:end]:
        t = l.strip()
        if not t or t.startswith(";"):
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            cur = m.group(1)
            blocks[cur] = []
            order.append(cur)
            continue
        if t.startswith("."):
            continue
        blocks[cur].append(t.split(";")[0].strip())
    nxt = {b: (order[i + 1] if i + 1 < len(order) else None) for i, b in enumerate(order)}
    rare = {b for b, ins in blocks.items() if any(i.startswith("v_fract_f64") for i in ins) and len(ins) < 12}

    best = {"cost": None}
    sys.setrecursionlimit(100000)

    def walk(block, idx, counts, phase, markers, decisions):
        # returns nothing; records the cheapest complete path in `best`
        while True:
            ins_list = blocks[block]
            if idx >= len(ins_list):
                block, idx = nxt[block], 0
                if block is None:
                    return
                continue
            ins = ins_list[idx]
            op = ins.split()[0]
            idx += 1
            if op in ("s_memtime", "s_memrealtime"):
                markers.append(op)
                if op == "s_memtime":
                    phase = sum(1 for m in markers if m == "s_memtime") - 1   # phase index after this stamp
                continue
            if op == "s_endpgm":
                if sum(1 for m in markers if m == "s_memtime") == 10:       # every stamp passed
                    cost = sum(v for (ph, c), v in counts.items() if c.startswith("v_"))
                    if best["cost"] is None or cost < best["cost"]:
                        best.update(cost=cost, counts=dict(counts), decisions=list(decisions))
                return
            if op.startswith("s_cbranch_exec"):
                target = ins.split()[1]
                counts[(phase, "salu")] += 1
                if op == "s_cbranch_execnz" and target not in rare:
                    block, idx = target, 0
                continue
            if op == "s_branch":
                counts[(phase, "salu")] += 1
                block, idx = ins.split()[1], 0
                continue
            if op.startswith("s_cbranch_"):
                counts[(phase, "salu")] += 1
                target = ins.split()[1]
                walk(target, 0, collections.Counter(counts), phase, list(markers), decisions + [(block, op, target, "taken")])
                decisions = decisions + [(block, op, target, "not taken")]
                continue
            counts[(phase, classify(ins))] += 1

    walk("entry", 0, collections.Counter(), -1, [], [])
    assert best["cost"] is not None, "no complete path found"
    counts = best["counts"]
    classes = ["v_fma64", "v_muladd64", "v_other", "mfma", "lds", "vmem", "salu", "smem", "wait"]
    print("static instruction budget per wavefront (= agent): fp64, SimpleCart, K = 10, T = 200, S = 4, no replay memory")
    print("(wavefront-uniform decisions on the counted path: %s)" % "; ".join(
        "%s %s -> %s %s" % d for d in best["decisions"]))
    print("%-52s" % "phase" + "".join("%11s" % c for c in classes) + "%11s" % "VALU all")
    tot = collections.Counter()
    for ph in range(-1, 10):
        row = [counts.get((ph, c), 0) for c in classes]
        if not any(row):
            continue
        name = "prologue (kernel arguments, lane -> steps)" if ph < 0 else (PHASES[ph] if ph < len(PHASES) else "epilogue")
        valu = sum(counts.get((ph, c), 0) for c in ("v_fma64", "v_muladd64", "v_other"))
        print("%-52s" % name + "".join("%11d" % v for v in row) + "%11d" % valu)
        for c, v in zip(classes, row):
            tot[c] += v
    valu = tot["v_fma64"] + tot["v_muladd64"] + tot["v_other"]
    print("%-52s" % "total" + "".join("%11d" % tot[c] for c in classes) + "%11d" % valu)
    print()
    print("vector-pipe cycles per agent: %d VALU x 4 + %d MFMA (4x4x4, 16 cycles each) x 16 = %d"
          % (valu, tot["mfma"], 4 * valu + 16 * tot["mfma"]))
    # the reference formulation's work at this shape: W = 2 K^2 N + 4 K^2 T + (4 K + 140) T flop
    K, T = 10, 200
    W = 2 * K * K * T + 4 * K * K * T + (4 * K + 140) * T
    print("reference-formulation work W = %d flop = %d fp64 multiply-adds = %.0f full 64-lane instructions; issued: %d "
          "fp64 FMA + %d fp64 mul/add + %d other vector + %d x 64 (MFMA: 256 multiply-adds per lane-group of 64) "
          % (W, W // 2, W / 2 / 64, tot["v_fma64"], tot["v_muladd64"], tot["v_other"], tot["mfma"]))
    print("lanes at T = 200: lanes 0..7 own 4 steps, lanes 8..63 three: the per-step instructions of slots 0..2 run on full "
          "wavefronts, those of the last slot on 8 lanes -- except its gradient, which all 64 lanes take together "
          "(8 lanes per step); 200 of 256 lane-steps carry a step (78 %)")


if __name__ == "__main__":
    main()
