#!/bin/bash
# where the compute stream waits: gaps between consecutive encoder launches of a complete shape and what ran in them
# (development aid).   WC_ARGS="p2s_max 4" bash tools/gap_timeline.sh
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/gaps
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$ROOT
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -o g -- python $ROOT/tools/vanilla_fixture.py ${WC_ARGS:-p2s_max 4} > $OUT/stdout.log 2>&1
python - <<PY > $OUT/gaps.txt
import csv, glob, collections
f = glob.glob('$OUT/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def name(r):
    return r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0].split('<')[0][:24]
ch = [r for r in rows if 'p2s_chain' in r['Kernel_Name']]
tot_gap, big = 0.0, []
for a, b in zip(ch[:-1], ch[1:]):
    g = (int(b['Start_Timestamp']) - int(a['End_Timestamp'])) / 1e6
    tot_gap += max(g, 0.0)
    if g > 1.5:
        busy = collections.Counter()
        for r in rows:
            s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
            if e > int(a['End_Timestamp']) and s < int(b['Start_Timestamp']) and 'p2s_chain' not in r['Kernel_Name']:
                busy[name(r)] += (min(e, int(b['Start_Timestamp'])) - max(s, int(a['End_Timestamp']))) / 1e6
        big.append((g, dict(busy.most_common(5))))
span = (int(ch[-1]['End_Timestamp']) - int(ch[0]['Start_Timestamp'])) / 1e6
busy_ch = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in ch) / 1e6
print('encoder launches %d, span %.1f ms, encoder busy %.1f ms, gaps %.1f ms' % (len(ch), span, busy_ch, tot_gap))
for g, b in big:
    print('gap %.2f ms: %s' % (g, b))
PY
find $OUT -name "*kernel_trace.csv" -delete
tail -1 $OUT/stdout.log; cat $OUT/gaps.txt | head -40
