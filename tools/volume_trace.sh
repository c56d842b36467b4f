#!/bin/bash
# per-launch durations and gaps of the sweep kernel (development aid): tools/volume_trace.sh [res]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/voltrace
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$ROOT
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT -o vol -- python $ROOT/tools/volume_bench.py ${1:-256} > $OUT/stdout.log 2>&1
python - <<PY
import csv, glob
f = glob.glob('$OUT/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'vol_sweep' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
n = len(rows) // 4
last = rows[-n:]
d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in last]
g = [(int(last[i + 1]['Start_Timestamp']) - int(last[i]['End_Timestamp'])) / 1e3 for i in range(n - 1)]
print('launches', n, 'sum_dur_us', round(sum(d), 1), 'sum_gap_us', round(sum(g), 1))
print('dur', ' '.join('%.1f' % x for x in d))
print('gap', ' '.join('%.1f' % x for x in g))
PY
