#!/bin/bash
# usage (on the GPU box, from the repo root):
#   tools/prof.sh <tag> [bench.py args...]                 profile bench.py (headline kernels)
#   tools/prof.sh <tag> --script tools/x.py [args...]      profile another python driver (configs[4], e2e loop ...)
# rocprofv3 kernel-trace stats + four PMC passes (separate runs, as MI355X_MICROARCH.md prescribes), CSV summaries into
# gpurun_out/<tag>/; the pieces worth committing are copied to gpurun_out/<tag>/commit/ (-> profiles/).
# Fails (exit 1, nothing written to commit/) when any of the five profiler runs fails or leaves no fresh output: the
# source-hash stamp bench.py trusts is written only beside counters that this very run produced.
set -u
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
rm -rf $OUT/trace $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 $OUT/pmc4 $OUT/commit $OUT/pmc.json     # never summarise an earlier revision's leftovers
mkdir -p $OUT
if [ "${1:-}" == "--script" ]; then
  SCRIPT=$GRAFT_REPO_ROOT/$2; shift 2
  TRACE_ARGS=("$@"); PMC_ARGS=("$@")
else
  SCRIPT=$GRAFT_REPO_ROOT/bench.py
  TRACE_ARGS=(--steps 10 --warmup 3 --no-cpu-baseline "$@"); PMC_ARGS=(--steps 4 --warmup 1 --no-cpu-baseline "$@")
fi
cd /tmp && export TMPDIR=/tmp
fail() { echo "tools/prof.sh $TAG: $1 -- no profile set written" >&2; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $SCRIPT "${TRACE_ARGS[@]}" > $OUT/trace_bench.log 2>&1 \
  || fail "kernel-trace run failed (see $OUT/trace_bench.log)"
ls $OUT/trace/*kernel_stats.csv > /dev/null 2>&1 || ls $OUT/trace/*/*kernel_stats.csv > /dev/null 2>&1 || fail "kernel-trace run left no kernel_stats.csv"
i=0
for PMC in "GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $OUT/pmc$i -o pmc -- python3 $SCRIPT "${PMC_ARGS[@]}" > $OUT/pmc${i}_bench.log 2>&1 \
    || fail "PMC pass $i failed (see $OUT/pmc${i}_bench.log)"
  [ -n "$(find $OUT/pmc$i -name '*counter_collection.csv' 2>/dev/null | head -1)" ] || fail "PMC pass $i left no counter_collection.csv"
done
python3 $GRAFT_REPO_ROOT/tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1 || fail "summarize_prof.py failed"
[ -s $OUT/pmc.json ] || fail "no pmc.json"
cat $OUT/summary.txt
mkdir -p $OUT/commit
cp $OUT/summary.txt $OUT/commit/${TAG}_rocprofv3_summary.txt
cp $OUT/pmc.json $OUT/commit/${TAG}_pmc.json
cp $(find $OUT/trace -name '*kernel_stats.csv' | head -1) $OUT/commit/${TAG}_kernel_stats.csv
# the bench line of the traced run: JSON for bench.py, the last text lines for the other drivers
if grep -q "^{" $OUT/trace_bench.log; then grep "^{" $OUT/trace_bench.log | tail -1 > $OUT/commit/${TAG}_bench_under_trace.json
else grep -v "amdgpu.ids\|rocprofv3\|^[EWI][0-9]\{8\}\|^$" $OUT/trace_bench.log | tail -4 > $OUT/commit/${TAG}_under_trace.txt; fi
# LAST: the kernel sources AND the library this set was taken from.  bench.py quotes a committed profile only while the
# first token still matches its own csrc_hash(); the second is the sha1 of the libkws_amd.so the profiled process loaded.
python3 -c "import sys, hashlib; sys.path.insert(0, '$GRAFT_REPO_ROOT'); import bench; from keyword_spotting_amd import _lib; print(bench.csrc_hash(), hashlib.sha1(open(_lib.LIB_PATH, 'rb').read()).hexdigest()[:16])" > $OUT/commit/${TAG}_source_hash.txt \
  || fail "could not stamp the source hash"
# the raw traces are large (the serving run alone leaves > 100 MB of kernel records) and gpurun copies back 64 MiB at most:
# only the summaries travel
rm -rf $OUT/trace $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 $OUT/pmc4
