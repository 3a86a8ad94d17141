#!/bin/bash
# A/B of the streams handled per gate workgroup in the fused gate + front-end launch (KWS_FE_GATE_STREAMS): kernel time under the
# tracer and HBM read bytes (FETCH_SIZE pass).  usage (GPU box): tools/exp_gate_ab.sh 8 16 32
cd /tmp && export TMPDIR=/tmp
for g in "$@"; do
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/gate_$g
  KWS_FE_GATE_STREAMS=$g rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/gate_$g/t -o trace -- python3 $GRAFT_REPO_ROOT/tools/bench_e2e.py > /dev/null 2>&1
  KWS_FE_GATE_STREAMS=$g rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/gate_$g/p -o pmc -- python3 $GRAFT_REPO_ROOT/tools/bench_e2e.py > /dev/null 2>&1
  t=$(grep -h 'mel_fft400_kernel<3, float, true' $GRAFT_REPO_ROOT/gpurun_out/gate_$g/t/trace_kernel_stats.csv | cut -d, -f5)
  f=$(python3 - <<PY
import csv,glob
v=[float(r["Counter_Value"]) for f in glob.glob("$GRAFT_REPO_ROOT/gpurun_out/gate_$g/p/*counter_collection.csv") for r in csv.DictReader(open(f)) if "float, true" in r["Kernel_Name"] and r["Counter_Name"]=="FETCH_SIZE"]
print("%.1f" % (sum(v)/len(v)*1024*2/1e6))
PY
)
  echo "gate streams per block $g: avg ns $t, HBM read MB (FETCH_SIZE x2) $f"
done
