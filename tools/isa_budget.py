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
Usage: python tools/isa_budget.py [--case metric|k20f32|k20f64] [--keep /tmp/budget.s]
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


CASES = {
    # name: (real type, model id, K, T, kernel symbol prefix, lean, WPB, label)
    "metric": ("double", 1, 10, 200, "_ZN3eea4wave19control_wave_kernelIdLi1ELi10ELb0ELi4E", False, 4,
               "fp64, SimpleCart, K = 10, T = 200"),
    "t192": ("double", 1, 10, 192, "_ZN3eea4wave19control_wave_kernelIdLi1ELi10ELb0ELi4E", False, 4,
             "fp64, SimpleCart, K = 10, T = 192 (three full slots: what the 8-step tail of T = 200 adds)"),
    "k20f32": ("float", 0, 20, 250, "_ZN3eea4wave19control_wave_kernelIfLi0ELi20ELb0ELi4E", False, 4,
               "fp32, Omni, K = 20, T = 250 (BASELINE configs[2])"),
    "k20f64": ("double", 0, 20, 250, "_ZN3eea4wave19control_wave_kernelIdLi0ELi20ELb0ELi1E", False, 1,
               "fp64, Omni, K = 20, T = 250 (configs[2] fp64 twin)"),
}


def classify(ins):
    op = ins.split()[0]
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_pk_fma_f32"):
        return "v_pkfma32"
    if op in ("v_fma_f64", "v_fmac_f64_e32", "v_fmac_f64_e64", "v_fmac_f64", "v_fma_f32", "v_fmac_f32_e32", "v_fmac_f32_e64",
              "v_fmac_f32", "v_fmac_f32_dpp", "v_fmac_f64_dpp"):
        return "v_fma64"
    if (op.startswith("v_mul_f64") or op.startswith("v_add_f64") or op.startswith("v_mul_f32") or op.startswith("v_add_f32")
            or op.startswith("v_sub_f32") or op.startswith("v_pk_mul_f32") or op.startswith("v_pk_add_f32")):
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
    case = sys.argv[sys.argv.index("--case") + 1] if "--case" in sys.argv else "metric"
    real, model, KC, TC, symbol, lean, wpb, label = CASES[case]
    tmp = tempfile.mkdtemp()
    out = keep or os.path.join(tmp, "budget.s")
    csrc = os.path.join(ROOT, "ergodic_exploration_amd", "csrc")
    # the kernel text with the case's shape as compile-time constants (one receding-horizon step per launch)
    text = open(os.path.join(csrc, "control_wave_impl.hpp")).read()
    subs = [("  int S = S_arg, rollout_only = rollout_arg;\n  asm volatile(\"\" : \"+s\"(S), \"+s\"(rollout_only));\n",
             "  constexpr int S = %d;\n  constexpr int rollout_only = 0;\n" % ((TC + 63) // 64)),
            ("  const int n_steps = RESIDENT ? 0x7fffffff : ((KernArgParams*)__builtin_amdgcn_kernarg_segment_ptr())->n_steps;\n",
             "  constexpr int n_steps = 1;\n"),
            ("  KernArgParams* ka = (KernArgParams*)__builtin_amdgcn_kernarg_segment_ptr();  // p_arg is argument 0\n"
             "  asm volatile(\"\" : \"+s\"(ka));\n  KernArgParams& p = *ka;\n",
             "  ControlParams<R> p = p_arg;\n  p.mem_cols = nullptr;\n  p.n_mem = nullptr;\n  p.mem_stride = 0;\n"
             "  p.traj = p.edx = p.bdx = p.rhot = nullptr;\n  p.ck = nullptr;\n  p.ck_rec = nullptr;\n  p.rec_ready = nullptr;\n"
             "  p.ck_shared = nullptr;\n  p.ck_shared_parts = 0;\n  p.ck_flag = nullptr;\n  p.done = nullptr;\n  p.K = %d;\n"
             "  p.pose_step_stride = p.u0_step_stride = 0;\n" % KC),
            ("  const int T = p.T;\n", "  constexpr int T = %d;\n" % TC)]
    for a, b in subs:
        assert text.count(a) == 1, a
        text = text.replace(a, b)
    with open(os.path.join(tmp, "budget_impl.hpp"), "w") as f:
        f.write(text)
    unit = os.path.join(tmp, "budget_unit.hip")
    with open(unit, "w") as f:
        f.write('#include "budget_impl.hpp"\nnamespace eea {\ntemplate __global__ void wave::%s<%s, %d, %d, false, %d, false>('
                "const ControlParams<%s>, const unsigned, const int, const int);\n}\n"
                % ("control_wave_kernel", real, model, KC, wpb, real))
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-function", "-fno-slp-vectorize",
           "-mllvm", "-disable-machine-licm",
           "-I", tmp, "-I", csrc, "-include", os.path.join(ROOT, "tools", "ab", "wave_stamps.hpp"), "-S", "--cuda-device-only",
           "-o", out, unit]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    src = open(out).read().split("\n")
    start = next(i for i, l in enumerate(src) if re.match(r"^" + symbol + r".*:", l))
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
            counts[(phase, "op:" + op)] += 1

    walk("entry", 0, collections.Counter(), -1, [], [])
    assert best["cost"] is not None, "no complete path found"
    counts = best["counts"]
    classes = ["v_fma64", "v_pkfma32", "v_muladd64", "v_other", "mfma", "lds", "vmem", "salu", "smem", "wait"]
    print("static instruction budget per wavefront (= agent): %s, S = %d, no replay memory, one step per launch" % (label, (TC + 63) // 64))
    print("(columns: v_fma64 = scalar FMA of the kernel's type, v_pkfma32 = packed fp32 FMA (2 per lane), v_muladd64 = mul / add)")
    print("(wavefront-uniform decisions on the counted path: %s)" % "; ".join(
        "%s %s -> %s %s" % d for d in best["decisions"]))
    print("%-52s" % "phase" + "".join("%11s" % c for c in classes) + "%11s" % "VALU all")
    tot = collections.Counter()
    for ph in range(-1, 10):
        row = [counts.get((ph, c), 0) for c in classes]
        if not any(row):
            continue
        name = "prologue (kernel arguments, lane -> steps)" if ph < 0 else (PHASES[ph] if ph < len(PHASES) else "epilogue")
        valu = sum(counts.get((ph, c), 0) for c in ("v_fma64", "v_pkfma32", "v_muladd64", "v_other"))
        print("%-52s" % name + "".join("%11d" % v for v in row) + "%11d" % valu)
        for c, v in zip(classes, row):
            tot[c] += v
    if "--opcodes" in sys.argv:   # the executed vector opcodes per phase (what "other" is made of)
        for ph in range(-1, 10):
            ops = sorted(((v, c[3:]) for (q, c), v in counts.items() if q == ph and c.startswith("op:v_")), reverse=True)
            if ops:
                name = "prologue" if ph < 0 else (PHASES[ph] if ph < len(PHASES) else "epilogue")
                print("   [%s] " % name + ", ".join("%s x%d" % (o, v) for v, o in ops))
    valu = tot["v_fma64"] + tot["v_pkfma32"] + tot["v_muladd64"] + tot["v_other"]
    print("%-52s" % "total" + "".join("%11d" % tot[c] for c in classes) + "%11d" % valu)
    print()
    # pipe time: fp64 16x16x4 64 cycles, fp64 4x4x4(4b) 16, fp32 4x4x1(16b) 8 (tools/ubench/mfma4x4x1.hip: the same
    # multiply-adds per cycle as the 16x16x4 form's 1024 in 32)
    mf = sum(v for (ph, c), v in counts.items() if c == "mfma")
    MF_CYCLES = {"v_mfma_f64_4x4x4_4b_f64": 16, "v_mfma_f64_16x16x4_f64": 64, "v_mfma_f32_4x4x1_16b_f32": 8,
                 "v_mfma_f32_16x16x4_f32": 32}
    mf_by_op = collections.Counter()
    for (ph, c), v in counts.items():
        if c.startswith("op:v_mfma"):
            mf_by_op[c[3:]] += v
    mf_total = sum(MF_CYCLES[o] * v for o, v in mf_by_op.items())
    print("vector-pipe cycles per agent: %d VALU x 4 + %s = %d" % (
        valu, " + ".join("%d %s x %d" % (v, o, MF_CYCLES[o]) for o, v in mf_by_op.items()) or "0 MFMA", 4 * valu + mf_total))
    # the reference formulation's work at this shape: W = 2 K^2 N + 4 K^2 T + (4 K + 140) T flop
    K, T = KC, TC
    W = 2 * K * K * T + 4 * K * K * T + (4 * K + 140) * T
    print("reference-formulation work W = %d flop = %d multiply-adds = %.0f full 64-lane instructions; issued: %d FMA + %d "
          "packed fp32 FMA + %d mul/add + %d other vector + %d MFMA"
          % (W, W // 2, W / 2 / 64, tot["v_fma64"], tot["v_pkfma32"], tot["v_muladd64"], tot["v_other"], tot["mfma"]))
    peak_note = ("fp32 vector peak 157.3 TF counts 2 multiply-adds per lane and cycle-slot (v_pk_fma_f32): a scalar fp32 FMA uses "
                 "half of it" if real == "float" else "fp64 vector peak 78.6 TF = one FMA per lane and 4-cycle issue")
    print(peak_note)


if __name__ == "__main__":
    main()
