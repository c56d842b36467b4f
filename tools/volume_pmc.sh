#!/bin/bash
# PMC pass (own run, kernel-trace only) for the sign-propagation sweep kernel: instruction mix and LDS conflicts
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/volpmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$ROOT
for grp in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_INSTS_SALU"; do
  name=$(echo $grp | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $grp -d $OUT/$name -o pmc -- python $ROOT/tools/volume_bench.py 256 > $OUT/$name.log 2>&1
done
python - <<PY
import csv, glob, json, collections
acc = collections.defaultdict(list)
dur = []
for f in glob.glob('$OUT/*/pmc_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'vol_sweep' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
            if r['Counter_Name'] == 'GRBM_GUI_ACTIVE':
                dur.append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
out = {k: {'dispatches': len(v), 'avg_per_dispatch': sum(v) / len(v)} for k, v in acc.items()}
out['avg_duration_us_under_pmc'] = sum(dur) / max(len(dur), 1) / 1e3
json.dump(out, open('$OUT/volume_pmc.json', 'w'), indent=1)
print(json.dumps(out))
PY
