#!/bin/bash
# timeline of every kernel of two consecutive chunks inside the full p2s_vanilla pipeline on the test shape (development aid)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/wctl
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$ROOT
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -o wc -- python $ROOT/tools/vanilla_fixture.py ${WC_ARGS:-} > $OUT/stdout.log 2>&1
python - <<PY > $OUT/timeline.txt
import csv, glob
f = glob.glob('$OUT/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
tabs = [i for i, r in enumerate(rows) if 'wc_tables' in r['Kernel_Name']]
i0, i1 = tabs[len(tabs) // 2], tabs[len(tabs) // 2 + 2]
t0 = int(rows[i0]['Start_Timestamp'])
for r in rows[i0:i1 + 1]:
    nm = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0].split('<')[0][:24]
    s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    if (e - s) < 20000 and 'gemm' in nm:
        continue
    print('%-24s q%-3s start %8.3f  end %8.3f  dur %7.3f ms' % (nm, r.get('Queue_Id', '?')[-3:], s / 1e6, e / 1e6, (e - s) / 1e6))
PY
python - <<PY > $OUT/periods.txt
import csv, glob, collections
f = glob.glob('$OUT/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def name(r):
    return r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0].split('<')[0][:24]
tabs = [i for i, r in enumerate(rows) if 'wc_tables' in r['Kernel_Name']]
per = []
for a, b in zip(tabs[:-1], tabs[1:]):
    dt = (int(rows[b]['Start_Timestamp']) - int(rows[a]['Start_Timestamp'])) / 1e6
    per.append(dt)
    if dt > 32.0:
        busy = collections.Counter()
        for r in rows[a:b]:
            busy[name(r)] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
        print('period %.1f ms at request %d:' % (dt, len(per)), ', '.join('%s %.1f' % kv for kv in busy.most_common(8)))
print('periods between wc_tables starts (ms):', ' '.join('%.0f' % p for p in per))
print('sum %.0f ms over %d periods, median %.1f' % (sum(per), len(per), sorted(per)[len(per) // 2]))
PY
find $OUT -name "*kernel_trace.csv" -delete
tail -1 $OUT/stdout.log
