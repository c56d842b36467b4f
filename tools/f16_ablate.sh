#!/bin/bash
# development aid (GPU box): timing-only ablations of the fp16-pair chain kernel (r03 schedule, P2S_F16_PIPE=0), built
# with tools/build_variant.sh abl<mask> -DP2S_F16_ABL=<mask>   (WRONG results: they leave work out)
out=${1:-gpurun_out/f16_ablate.txt}
mkdir -p $(dirname $out)
run() {
  label=$1; shift
  r=$(env "$@" python tools/quick_bench.py --model p2s_max --B 8192 --iters 3 --bf16 4 2>/dev/null | tail -1 | python -c '
import sys, json
d = json.loads(sys.stdin.read())
print("%.2f ms  chain_stn %.2f chain_main %.2f" % (d["ms"], d["stages_ms"]["ms_chain_stn"], d["stages_ms"]["ms_chain_main"]))')
  echo "$label: $r" | tee -a $out
}
run "baseline (r03 schedule)" P2S_F16_PIPE=0
for v in build_variants/abl*.so; do
  [ -f "$v" ] && run "$(basename $v .so)" P2S_F16_PIPE=0 P2S_LIB_PATH=$PWD/$v
done
