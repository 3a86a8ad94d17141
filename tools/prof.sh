#!/bin/bash
# usage (on the GPU box, from the repo root):
#   tools/prof.sh <tag> [bench.py args...]                 profile bench.py (headline kernels)
#   tools/prof.sh <tag> --script tools/x.py [args...]      profile another python driver (configs[4], e2e loop ...)
# rocprofv3 kernel-trace stats + four PMC passes (separate runs, as MI355X_MICROARCH.md prescribes), CSV summaries into
# gpurun_out/<tag>/; the pieces worth committing are copied to gpurun_out/<tag>/commit/ (-> profiles/).
set -u
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
if [ "${1:-}" == "--script" ]; then
  SCRIPT=$GRAFT_REPO_ROOT/$2; shift 2
  TRACE_ARGS=("$@"); PMC_ARGS=("$@")
else
  SCRIPT=$GRAFT_REPO_ROOT/bench.py
  TRACE_ARGS=(--steps 10 --warmup 3 --no-cpu-baseline "$@"); PMC_ARGS=(--steps 4 --warmup 1 --no-cpu-baseline "$@")
fi
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $SCRIPT "${TRACE_ARGS[@]}" > $OUT/trace_bench.log 2>&1
i=0
for PMC in "GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $OUT/pmc$i -o pmc -- python3 $SCRIPT "${PMC_ARGS[@]}" > $OUT/pmc${i}_bench.log 2>&1
done
python3 $GRAFT_REPO_ROOT/tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
mkdir -p $OUT/commit
cp $OUT/summary.txt $OUT/commit/${TAG}_rocprofv3_summary.txt
# the kernel sources this set was taken from: bench.py quotes a committed profile only while this still matches the build
python3 -c "import sys; sys.path.insert(0, '$GRAFT_REPO_ROOT'); import bench; print(bench.csrc_hash())" > $OUT/commit/${TAG}_source_hash.txt
cp $OUT/pmc.json $OUT/commit/${TAG}_pmc.json
cp $OUT/trace/*kernel_stats.csv $OUT/commit/${TAG}_kernel_stats.csv 2>/dev/null
# the bench line of the traced run: JSON for bench.py, the last text lines for the other drivers
if grep -q "^{" $OUT/trace_bench.log; then grep "^{" $OUT/trace_bench.log | tail -1 > $OUT/commit/${TAG}_bench_under_trace.json
else grep -v "amdgpu.ids\|rocprofv3\|^[EWI][0-9]\{8\}\|^$" $OUT/trace_bench.log | tail -4 > $OUT/commit/${TAG}_under_trace.txt; fi
