#!/bin/bash
# PMC passes (own runs) for the weighted sub-sample kernels alone (development aid)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/wcpmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$ROOT
for grp in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo $grp | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $grp -d $OUT/$name -o pmc -- python $ROOT/tools/vanilla_bench.py --skip-pipeline > $OUT/$name.log 2>&1
done
python - <<PY
import csv, glob, json, collections
acc = collections.defaultdict(list)
for f in glob.glob('$OUT/*/pmc_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        for kn in ('wc_tables', 'wc_spec', 'wc_ids'):
            if kn in r['Kernel_Name'] and int(r['Grid_Size']) >= 4096 * 256 // (8 if kn == 'wc_spec' else 1) // 4:
                acc[(kn, r['Counter_Name'])].append(float(r['Counter_Value']))
out = {'%s.%s' % k: round(sum(v) / len(v)) for k, v in sorted(acc.items())}
print(json.dumps(out))
PY
