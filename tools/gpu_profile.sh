#!/bin/bash
# rocprofv3 passes for the round's profile summaries (run on the GPU box via gpurun).
# usage: tools/gpu_profile.sh <tag>
TAG=${1:-r2}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$ROOT
# (1) kernel trace + stats of the bench command (no CPU baseline leg)
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python $ROOT/bench.py --steps 1 --warmup 1 --cpu-seconds 0 > $OUT/trace_stdout.log 2>&1
# (1b) the same for the secondary workloads: p2s_vanilla pipeline, split-bf16 encoder, sign propagation + iso-surface
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_vanilla -o vanilla -- python $ROOT/tools/vanilla_bench.py > $OUT/vanilla_stdout.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_bf16x3 -o bf16x3 -- python $ROOT/bench.py --steps 1 --warmup 1 --cpu-seconds 0 --bf16 3 > $OUT/bf16x3_stdout.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_fp16x2 -o fp16x2 -- python $ROOT/bench.py --steps 1 --warmup 1 --cpu-seconds 0 --bf16 4 --no-secondary > $OUT/fp16x2_stdout.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_volume -o volume -- python $ROOT/tools/volume_bench.py 256 > $OUT/volume_stdout.log 2>&1
if [ -n "$P2S_PROFILE_LIGHT" ]; then find $OUT -name "*.csv" | head -80 > $OUT/files.txt; exit 0; fi   # traces only (the PMC passes of an earlier run stay valid while the encoder kernels are unchanged)
# (2) PMC passes on a smaller encoder-only workload, kernel-trace only (counters in their own runs)
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo $grp | cut -d' ' -f1)
  timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc $grp -d $OUT/pmc_$name -o pmc -- python $ROOT/tools/quick_bench.py --B 4096 --iters 1 > $OUT/pmc_$name.log 2>&1
done
# (3) the same counters for the split-bf16 kernel (bf16 MFMA ops)
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo $grp | cut -d' ' -f1)
  timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc $grp -d $OUT/pmcbf_$name -o pmc -- python $ROOT/tools/quick_bench.py --B 4096 --iters 1 --bf16 3 > $OUT/pmcbf_$name.log 2>&1
done
# (4) and for the fp16 pair kernel
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo $grp | cut -d' ' -f1)
  timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc $grp -d $OUT/pmcf16_$name -o pmc -- python $ROOT/tools/quick_bench.py --B 4096 --iters 1 --bf16 4 > $OUT/pmcf16_$name.log 2>&1
done
rocprofv3 -L 2>/dev/null | grep -i -E "mfma|FETCH_SIZE|WRITE_SIZE|GRBM_GUI|LDS_BANK" | head -40 > $OUT/counters_available.txt
find $OUT -name "*.csv" | head -80 > $OUT/files.txt
du -sh $OUT >> $OUT/files.txt
