#!/bin/bash
# PMC passes (own runs, kernel-trace only) for the plain bf16 encoder kernel (one piece per operand; outside the accuracy
# contract, reported for completeness): tools/pmc_bf16.sh <tag>  ->  gpurun_out/prof_<tag>/pmcb1_*
TAG=${1:-r2}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$ROOT
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo $grp | cut -d' ' -f1)
  timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc $grp -d $OUT/pmcb1_$name -o pmc -- python $ROOT/tools/quick_bench.py --B 4096 --iters 1 --bf16 1 > $OUT/pmcb1_$name.log 2>&1
done
