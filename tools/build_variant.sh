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
# the product's own Makefile (flags, per-file options) on the patched copy; include/ sits two levels above csrc there too
mkdir -p $W/include && cp $ROOT/include/*.h $W/include/
# KWS_VARIANT_BUILD is defined HERE and nowhere else: csrc/kws_internal.h refuses an experiment switch without it, and
# kws_version() of the result says "VARIANT BUILD" and which switches
make -s -C $C -j8 OUT=$OUT/libkws_$NAME.so SPILL_CHECK= EXTRA="-DKWS_VARIANT_BUILD=1 ${FLAGS[*]}"
rm -rf $W
echo built $OUT/libkws_$NAME.so
