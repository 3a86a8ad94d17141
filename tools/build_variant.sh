#!/bin/bash
# tools/build_variant.sh <name> [--patch tools/patches/<x>.patch ...] [-DFLAG ...]  ->  variants/libkws_<name>.so
# Experiment builds (timing instrumentation, ablations) are made from a patched COPY of the sources, so the product
# sources carry no experiment code.  tools/patches/timing_ablation.patch re-adds the s_memtime phase counters
# (-DKWS_TIMING) and the KWS_ABL_* ablation switches as they were at the end of round 1 (it applies to that
# revision of the kernels; refresh it when the frame loop changes).  Load the result with KWS_AMD_LIB=variants/libkws_<name>.so.
set -e
NAME=$1; shift
ROOT=$(cd $(dirname $0)/.. && pwd)
OUT=$ROOT/variants; mkdir -p $OUT
W=$(mktemp -d)
mkdir -p $W/keyword_spotting_amd && cp -r $ROOT/keyword_spotting_amd/csrc $W/keyword_spotting_amd/csrc && rm -rf $W/keyword_spotting_amd/csrc/_obj
FLAGS=()
while [ $# -gt 0 ]; do
  if [ "$1" == "--patch" ]; then P=$(realpath $2); (cd $W && patch -s -p1 < $P); shift 2; else FLAGS+=("$1"); shift; fi
done
C=$W/keyword_spotting_amd/csrc
# per-file flags as in csrc/Makefile (FLAGS_<file>)
OBJ=$W/obj; mkdir -p $OBJ
for f in $C/*.hip; do
  n=$(basename $f .hip); X=""
  [ "$n" == "gru_bf16" ] && X="-mllvm -amdgpu-mfma-vgpr-form=1"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include -I$C -Wno-unused-value -Wno-unused-result $X "${FLAGS[@]}" -c $f -o $OBJ/$n.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJ/*.o -o $OUT/libkws_$NAME.so
rm -rf $W
echo built $OUT/libkws_$NAME.so
