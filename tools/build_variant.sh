#!/bin/bash
# development aid: build libp2s_hip.so variants with extra -D flags into build_variants/<name>.so (select with P2S_LIB_PATH)
#   tools/build_variant.sh hold0 -DP2S_BF16_HOLD=0
name=$1; shift
ROOT=$(cd $(dirname $0)/.. && pwd)
mkdir -p $ROOT/build_variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-gpu-rdc -Wall -Wno-unused-function "$@" \
  -o $ROOT/build_variants/$name.so $ROOT/points2surf_amd/csrc/*.hip
