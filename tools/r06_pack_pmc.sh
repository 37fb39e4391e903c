#!/bin/bash
# SQ counters of the packed kernel at the two chip-filling short-horizon shapes ("commit the counter table
# that shows why"): separate rocprofv3 --pmc passes (no trace domains beside --kernel-trace), one stand-alone leg each.
# Run on the GPU box: tools/r06_pack_pmc.sh [tag] -> gpurun_out/pack_pmc_<tag>/summary.txt
set -u
TAG=${1:-r06}
OUT=gpurun_out/pack_pmc_${TAG}
mkdir -p "$OUT"
export TMPDIR=/tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY"
P2="SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM GRBM_GUI_ACTIVE"
i=0
while IFS='|' read -r CASE EXTRA; do
  i=$((i + 1))
  echo "$CASE|$EXTRA" > "$OUT/c$i.case"
  rocprofv3 --kernel-trace --pmc $P1 --kernel-include-regex control_ --output-format csv -d "$OUT/c${i}_p1" -o pmc -- python3 tools/other_config_point.py --case "$CASE" --spinup-s 0 $EXTRA > "$OUT/c${i}_p1.log" 2>&1
  rocprofv3 --kernel-trace --pmc $P2 --kernel-include-regex control_ --output-format csv -d "$OUT/c${i}_p2" -o pmc -- python3 tools/other_config_point.py --case "$CASE" --spinup-s 0 $EXTRA > "$OUT/c${i}_p2.log" 2>&1
done <<'CASES'
explore_omni.yaml as shipped, chip-filling batch|
configs[1], chip-filling batch|
configs[3] with the Omni model|
CASES
python3 - "$OUT" <<'PY' > "$OUT/summary.txt" 2>&1
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
for casef in sorted(glob.glob(os.path.join(out, "c*.case"))):
    tag = casef[:-5]
    name = open(casef).read().strip()
    c = defaultdict(list)
    res = None
    for p in glob.glob(tag + "_p*/**/*counter_collection.csv", recursive=True):
        with open(p, newline="") as f:
            for r in csv.DictReader(f):
                if "control_" in r.get("Kernel_Name", ""):
                    c[r["Counter_Name"]].append(float(r["Counter_Value"]))
                    res = res or (r["Kernel_Name"].split("(")[0][:110], r.get("VGPR_Count"), r.get("LDS_Block_Size"), r.get("Grid_Size"), r.get("Workgroup_Size"))
    m = {k: sum(v) / len(v) for k, v in c.items()}
    print("== %s" % name)
    print("   kernel %s  VGPR_Count %s  LDS %s  grid %s  workgroup %s  (%d dispatches per counter)" % (res + (max(len(v) for v in c.values()),)))
    for k in sorted(m):
        print("   %-32s %.6g" % (k, m[k]))
    w = m.get("SQ_WAVES", 0.0)
    if w:
        print("   per wavefront: VALU %.1f  (of which matrix ops %.1f)  LDS %.1f  SALU %.1f  SMEM %.1f" % (
            m.get("SQ_INSTS_VALU", 0) / w, m.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0) / w / 64.0 if False else m.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0) / w,
            m.get("SQ_INSTS_LDS", 0) / w, m.get("SQ_INSTS_SALU", 0) / w, m.get("SQ_INSTS_SMEM", 0) / w))
    wc = m.get("SQ_WAVE_CYCLES", 0.0)
    if wc:
        print("   SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES   %.3f   (share of wavefront time spent waiting for an instruction's operands / issue)" % (m.get("SQ_WAIT_INST_ANY", 0) / wc))
        print("   SQ_WAIT_ANY / SQ_WAVE_CYCLES        %.3f" % (m.get("SQ_WAIT_ANY", 0) / wc))
        print("   SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES %.3f  SQ_ACTIVE_INST_ANY / SQ_BUSY_CYCLES %.3f" % (
            m.get("SQ_ACTIVE_INST_VALU", 0) / max(m.get("SQ_BUSY_CYCLES", 1), 1), m.get("SQ_ACTIVE_INST_ANY", 0) / max(m.get("SQ_BUSY_CYCLES", 1), 1)))
        print("   LDS bank conflict cycles / LDS instructions %.3f" % (m.get("SQ_LDS_BANK_CONFLICT", 0) / max(m.get("SQ_INSTS_LDS", 1), 1)))
PY
cat "$OUT/summary.txt"
