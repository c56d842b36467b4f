#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_wchoice.py "tests/test_gpu_stress.py" -k "weighted or wchoice or speculative or vanilla" -q > $O/t9_wchoice.log 2>&1
echo "rc=$?" >> $O/t9_wchoice.log
P2S_WC_STATS=1 python tools/skip_bench.py --models p2s_vanilla --skip-only --reps 1 > $O/skip9_stats.json 2> $O/skip9_stats.err
python tools/skip_bench.py --models p2s_vanilla > $O/skip9.json 2> $O/skip9.err
P2S_WC_NO_OVERLAP=1 python tools/skip_bench.py --models p2s_vanilla --skip-only > $O/skip9_noov.json 2>> $O/skip9.err
python tools/skip_bench.py --models p2s_vanilla --shape 0 --skip-only > $O/skip9_shape0.json 2>> $O/skip9.err
timeout 900 python -m pytest tests/test_gpu_sizes.py -k "vanilla and (256 or 64 or 32 or 128)" -x -q > $O/t9_sizes.log 2>&1
echo "rc=$?" >> $O/t9_sizes.log
