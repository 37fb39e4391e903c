set -u
export TMPDIR=/tmp
python -m pytest tests/test_gpu_packed_agents.py -x -q 2>&1 | tail -5
python tools/pack_sweep.py --spl 50 --batches 4096,12288,24576,36864,49152 --out gpurun_out/pack_sweep_v2.json 2>&1 | tail -70
mkdir -p gpurun_out/pack_pmc
for cfg in "1 8 24576" "2 16 12288" "1 64 4096" "2 64 4096"; do
  set -- $cfg
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY --kernel-include-regex control_ --output-format csv -d gpurun_out/pack_pmc/s$1_l$2 -o pmc -- python3 tools/pack_point.py --shape $1 --lanes $2 --agents $3 > gpurun_out/pack_pmc/s$1_l$2.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-include-regex control_ --output-format csv -d gpurun_out/pack_pmc/s$1_l$2b -o pmc -- python3 tools/pack_point.py --shape $1 --lanes $2 --agents $3 > gpurun_out/pack_pmc/s$1_l$2b.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/pack_pmc/s*_l*/')):
    acc = collections.defaultdict(list)
    for p in glob.glob(d + '**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(p)):
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    print(d, {k: sum(v)/len(v) for k, v in acc.items()})
PY
