#!/bin/bash
# PMC passes (own runs, kernel trace only) of the encoder kernels behind profiles/r05/pmc_summary.json: fp32 chain kernel and
# the fp16-pair kernel (incl. cache-side counters of the latter).  tools/pmc_chain_r05.sh  ->  gpurun_out/prof_r05/pmc*_*
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_r05
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$ROOT
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo $grp | cut -d' ' -f1)
  timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc $grp -d $OUT/pmc_$name -o pmc -- python $ROOT/tools/quick_bench.py --B 4096 --iters 1 > $OUT/pmc_$name.log 2>&1
done
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM"; do
  name=$(echo $grp | cut -d' ' -f1)
  timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc $grp -d $OUT/pmcf16_$name -o pmc -- python $ROOT/tools/quick_bench.py --B 4096 --iters 1 --bf16 4 > $OUT/pmcf16_$name.log 2>&1
done
