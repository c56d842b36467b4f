#!/bin/bash
# rocprofv3 passes behind profiles/r05 (run on the GPU box via gpurun): tools/gpu_profile_r05.sh
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_r05
mkdir -p $OUT
cd $ROOT
python bench.py --steps 20 --warmup 5 > $OUT/bench_r05.json 2> $OUT/bench_r05.err
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$ROOT
# kernel traces + stats: the headline command, configs[3]'s model in fp32 and with the fp16-pair encoder, the stream skip alone
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python $ROOT/bench.py --steps 1 --warmup 1 --cpu-seconds 0 --no-secondary > $OUT/bench_profiled.json 2> $OUT/trace.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_vanilla -o vanilla -- python $ROOT/bench.py --model p2s_vanilla --steps 1 --warmup 1 --cpu-seconds 0 > $OUT/vanilla_profiled.json 2> $OUT/trace_vanilla.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_vanilla_fp16x2 -o vanilla_fp16x2 -- python $ROOT/bench.py --model p2s_vanilla --bf16 4 --steps 1 --warmup 1 --cpu-seconds 0 > $OUT/vanilla_fp16x2_profiled.json 2> $OUT/trace_vanilla_fp16x2.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_skip -o skip -- python $ROOT/tools/skip_bench.py --models p2s_vanilla --skip-only --reps 2 > $OUT/skip_profiled.json 2> $OUT/trace_skip.err
# counters of the stream skip's kernels, in their own passes (kernel trace only)
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo $grp | cut -d' ' -f1)
  timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc $grp -d $OUT/pmcskip_$name -o pmc -- python $ROOT/tools/skip_bench.py --models p2s_vanilla --skip-only --reps 1 --res 128 > $OUT/pmcskip_$name.log 2>&1
done
python $ROOT/tools/skip_bench.py > $OUT/skip_bench.json 2> $OUT/skip_bench.err
python $ROOT/tools/skip_bench.py --models p2s_vanilla --encoder 4 > $OUT/skip_bench_fp16x2.json 2>> $OUT/skip_bench.err
find $OUT -name "*.csv" | head -80 > $OUT/files.txt
du -sh $OUT >> $OUT/files.txt
