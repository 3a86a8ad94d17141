#!/bin/bash
# tools/build_variant.sh <name> [-DFLAG ...]  -> gpurun_variants/libkws_<name>.so (timing experiments)
set -e
NAME=$1; shift
ROOT=$(cd $(dirname $0)/.. && pwd)
OUT=$ROOT/variants; mkdir -p $OUT
C=$ROOT/keyword_spotting_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I$ROOT/include -I$C -Wno-unused-value -Wno-unused-result "$@" \
  $C/gru_kernels.hip $C/gru_bf16.hip $C/gru_octbit.hip $C/frontend_kernels.hip $C/stream_kernels.hip $C/decode_kernels.hip $C/octbit_kernels.hip $C/kws_api.hip -o $OUT/libkws_$NAME.so
echo built $OUT/libkws_$NAME.so
