set -u
cd $GRAFT_REPO_ROOT
tools/prof.sh r4 > /dev/null 2>&1
tools/prof.sh r4_f16x3 --precision f16x3 > /dev/null 2>&1
tools/prof.sh r4_bf16 --precision bf16 > /dev/null 2>&1
tools/prof.sh r4_i8 --precision int8 > /dev/null 2>&1
tools/prof.sh r4_e2e --script tools/bench_e2e.py > /dev/null 2>&1
tools/prof.sh r4_e2ef16 --script tools/bench_e2e.py --precision f16x3 > /dev/null 2>&1
tools/prof.sh r4_fe --script tools/bench_fe.py > /dev/null 2>&1
tools/prof.sh r4_configC --script tools/bench_config.py > /dev/null 2>&1
ls gpurun_out/*/commit | head -60
