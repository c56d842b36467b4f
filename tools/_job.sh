#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O
cd $R
P2S_WC_STATS=1 python tools/skip_bench.py --models p2s_vanilla --skip-only --reps 1 --shape 0 > $O/skip16_s0.json 2> $O/skip16_s0.err
P2S_WC_STATS=1 python tools/skip_bench.py --models p2s_vanilla --skip-only --reps 1 --shape 1 > $O/skip16_s1.json 2> $O/skip16_s1.err
