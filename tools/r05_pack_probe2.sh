set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/pack_pmc2
for cfg in "8 8 3072" "8 16 3072" "8 24 3072" "8 32 3072" "16 16 3072" "16 32 3072" "16 48 3072" "16 64 3072" "64 64 3072" "64 128 3072" "64 192 3072"; do
  set -- $cfg
  A=$((64 / $1))
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_LDS_IDX_ACTIVE --kernel-include-regex control_ --output-format csv -d gpurun_out/pack_pmc2/l$1_t$2 -o pmc -- python3 tools/pack_point.py --shape 1 --lanes $1 --agents $(($3 * A)) --steps $2 --launches 4 > gpurun_out/pack_pmc2/l$1_t$2.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/pack_pmc2/l*_t*/')):
    acc = collections.defaultdict(list)
    for p in glob.glob(d + '**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(p)):
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    w = sum(acc['SQ_WAVES'])/len(acc['SQ_WAVES'])
    print(d, {k: round(sum(v)/len(v)/w, 1) for k, v in acc.items()})
PY
