#!/bin/bash
# The whole profile set of a round, once, at the end: tools/prof_all.sh r6   (on the GPU box, from the repo root).
# Stops at the first profile that fails (tools/prof.sh exits non-zero and writes no stamp in that case).
set -eu
R=${1:-r6}
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; tools/prof.sh "$@" > gpurun_out/prof_$1.log 2>&1 || { tail -5 gpurun_out/prof_$1.log; exit 1; }; }
mkdir -p gpurun_out
run $R
run ${R}_f16x3 --precision f16x3
run ${R}_bf16 --precision bf16
run ${R}_i8 --precision int8
run ${R}_e2e --script tools/bench_e2e.py
run ${R}_e2ef16 --script tools/bench_e2e.py --precision f16x3
run ${R}_e2ebf16 --script tools/bench_e2e.py --precision bf16
run ${R}_fe --script tools/bench_fe.py
run ${R}_configC --script tools/bench_config.py
run ${R}_configCf16 --script tools/bench_config.py --precision f16x3
run ${R}_serve --script tools/bench_serve.py --precision f16x3 --periods 8 --json
ls gpurun_out/*/commit | head -80
