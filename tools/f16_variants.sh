#!/bin/bash
# development aid (GPU box): A/B of the fp16-pair chain kernel schedules.  Variants are built beforehand with
#   tools/build_variant.sh f16wg2 -DP2S_F16_WG=2        (-> build_variants/f16wg2.so, selected with P2S_LIB_PATH)
# usage: tools/f16_variants.sh [out-file]
out=${1:-gpurun_out/f16_variants.txt}
mkdir -p $(dirname $out)
run() {  # label, env...
  label=$1; shift
  for model in p2s_max p2s_vanilla; do
    r=$(env "$@" python tools/quick_bench.py --model $model --B 8192 --iters 3 --bf16 4 2>/dev/null | tail -1 | python -c '
import sys, json
d = json.loads(sys.stdin.read())
print("%.2f ms  %.0f q/s  chain_stn %.2f chain_main %.2f head %.2f dec %.2f" % (d["ms"], d["qps"], d["stages_ms"]["ms_chain_stn"], d["stages_ms"]["ms_chain_main"], d["stages_ms"]["ms_stn_head"], d["stages_ms"]["ms_decoder"]))')
    echo "$label $model: $r" | tee -a $out
  done
}
run "r03-schedule     " P2S_F16_PIPE=0
run "pipelined wg3    " P2S_F16_PIPE=1
for v in build_variants/*.so; do
  [ -f "$v" ] && run "pipelined $(basename $v .so)" P2S_F16_PIPE=1 P2S_LIB_PATH=$PWD/$v
done
