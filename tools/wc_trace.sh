#!/bin/bash
# kernel stats of the weighted sub-sample stage alone (development aid)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/wctrace
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o wc -- python $ROOT/tools/vanilla_bench.py --skip-pipeline > $OUT/stdout.log 2>&1
python - <<PY
import csv, glob
f = glob.glob('$OUT/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print(r['Name'][:60].ljust(60), r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
PY
