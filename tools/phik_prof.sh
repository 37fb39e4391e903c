#!/bin/bash
# per-kernel durations of the phi_k path on large grids (rocprofv3 kernel trace)
export TMPDIR=/tmp
OUT=gpurun_out/prof_phik
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 tools/phik_bench.py > $OUT/run.log 2>&1
grep grid $OUT/run.log
python3 - <<'PY'
import csv,glob
seen=set()
for p in glob.glob('gpurun_out/prof_phik/**/*kernel_trace.csv',recursive=True):
    rows=list(csv.DictReader(open(p)))
    for r in rows:
        n=r['Kernel_Name']
        if 'spatial_pass1' in n:
            key=(n.split('(')[0][-40:], r['Grid_Size_X'], r['Grid_Size_Y'])
            us=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
            if key not in seen:
                seen.add(key)
            print(key, 'us', us)
PY
