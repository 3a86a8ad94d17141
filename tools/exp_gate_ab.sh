cd /tmp && export TMPDIR=/tmp
for g in 16 24 32 48; do
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/r4s_$g
  KWS_FE_GATE_STREAMS=$g rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4s_$g -o trace -- python3 $GRAFT_REPO_ROOT/tools/bench_e2e.py > /dev/null 2>&1
  echo "gate streams per block $g: $(grep -h 'mel_fft400_kernel<3, float, true' $GRAFT_REPO_ROOT/gpurun_out/r4s_$g/trace_kernel_stats.csv | cut -d, -f2-5)"
done
