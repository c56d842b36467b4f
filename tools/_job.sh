#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_wchoice.py "tests/test_gpu_stress.py" -k "weighted or wchoice or speculative or vanilla" -q > $O/t21_wchoice.log 2>&1
echo "rc=$?" >> $O/t21_wchoice.log
python tools/skip_bench.py --models p2s_vanilla > $O/skip21.json 2> $O/skip21.err
python tools/skip_bench.py --models p2s_vanilla --shape 0 > $O/skip21_shape0.json 2>> $O/skip21.err
python tools/skip_bench.py --models p2s_vanilla --encoder 4 > $O/skip21_f16.json 2>> $O/skip21.err
timeout 900 python -m pytest tests/test_gpu_sizes.py -k "vanilla and (256 or 64 or 32 or 128)" -x -q > $O/t21_sizes.log 2>&1
echo "rc=$?" >> $O/t21_sizes.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_skip21 -o skip -- python $R/tools/skip_bench.py --models p2s_vanilla --skip-only --reps 2 > $O/trace_skip21.log 2>&1
