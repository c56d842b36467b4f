#!/bin/bash
# timeline of the weighted sub-sample kernels of one chunk inside the full p2s_vanilla pipeline (development aid)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/wctl
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$ROOT
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT -o wc -- python $ROOT/tools/vanilla_bench.py > $OUT/stdout.log 2>&1
python - <<PY
import csv, glob
f = glob.glob('$OUT/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
tabs = [i for i, r in enumerate(rows) if 'wc_tables' in r['Kernel_Name']]
i0, i1 = tabs[len(tabs) // 2], tabs[len(tabs) // 2 + 1]
t0 = int(rows[i0]['Start_Timestamp'])
for r in rows[i0:i1 + 1]:
    nm = r['Kernel_Name'].split('(')[0].split('::')[-1][:28]
    s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    print('%-28s start %9.3f ms  dur %8.3f ms' % (nm, s / 1e6, (e - s) / 1e6))
PY
