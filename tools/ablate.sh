#!/bin/bash
# timing-only ablations of the chain kernel (development aid).  Needs a library built with the ablation switch:
#   P2S_EXTRA_HIPCC_FLAGS=-DP2S_DEV_ABLATE python -m points2surf_amd.build --force
run() {
  python tools/quick_bench.py --B 4096 --iters 3 2>/dev/null | tail -1 | python -c '
import sys, json
d = json.loads(sys.stdin.read())
print(round(d["ms"], 2), "ms; chain_stn", round(d["stages_ms"]["ms_chain_stn"], 2), "chain_main", round(d["stages_ms"]["ms_chain_main"], 2))'
}
for ab in ${ABL:-0 1 2}; do
  echo -n "ablate=$ab: "
  P2S_CHAIN_ABLATE=$ab run
done
