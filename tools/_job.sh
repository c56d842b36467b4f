#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O
cd $R
P2S_WC_STATS=1 python tools/skip_bench.py --models p2s_vanilla --skip-only --reps 1 > $O/skip13_stats.json 2> $O/skip13_stats.err
python tools/skip_bench.py --models p2s_vanilla > $O/skip13.json 2> $O/skip13.err
python tools/skip_bench.py --models p2s_vanilla --shape 0 --skip-only > $O/skip13_shape0.json 2>> $O/skip13.err
timeout 1700 python -m pytest tests -m gpu -q > $O/t13_all.log 2>&1
echo "rc=$?" >> $O/t13_all.log
python bench.py --steps 2 --warmup 1 > $O/bench13.json 2> $O/bench13.err
echo "rc=$?" >> $O/bench13.err
