#!/bin/bash
# rocprofv3 passes behind profiles/r06 (run on the GPU box via gpurun): tools/gpu_profile_r06.sh [quick]
#   unprofiled headline with the driver's flags; kernel trace + stats of the headline command; separate --pmc passes
#   (kernel trace only, as MI355X_MICROARCH.md prescribes) of the fp32 chain kernel with the 16-row tail tile;
#   bench.py --model p2s_vanilla [--bf16 4] (configs[3], incl. the data-path self-check); the stream skip
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_r06
mkdir -p $OUT
cd $ROOT
python bench.py --steps 20 --warmup 5 > $OUT/bench_r06.json 2> $OUT/bench_r06.err
python bench.py --model p2s_vanilla --steps 3 --warmup 1 --cpu-seconds 0 > $OUT/vanilla_r06.json 2> $OUT/vanilla_r06.err
python bench.py --model p2s_vanilla --bf16 4 --steps 3 --warmup 1 --cpu-seconds 0 > $OUT/vanilla_fp16x2_r06.json 2> $OUT/vanilla_fp16x2_r06.err
python tools/skip_bench.py --models p2s_vanilla > $OUT/skip_bench.json 2> $OUT/skip_bench.err
python tools/skip_bench.py --models p2s_vanilla --encoder 4 > $OUT/skip_bench_fp16x2.json 2>> $OUT/skip_bench.err
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$ROOT
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python $ROOT/bench.py --steps 1 --warmup 1 --cpu-seconds 0 --no-secondary > $OUT/trace_stdout.log 2> $OUT/trace.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_vanilla -o vanilla -- python $ROOT/tools/vanilla_fixture.py > $OUT/vanilla_fixture_profiled.log 2> $OUT/trace_vanilla.err
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo $grp | cut -d' ' -f1)
  timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc $grp -d $OUT/pmc_$name -o pmc -- python $ROOT/tools/quick_bench.py --B 4096 --iters 1 > $OUT/pmc_$name.log 2>&1
done
find $OUT -name "*.csv" | head -80 > $OUT/files.txt
find $OUT -name "*kernel_trace.csv" -size +20M -delete
du -sh $OUT >> $OUT/files.txt
