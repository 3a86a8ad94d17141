#!/bin/bash
# runs bench.py once per variant library, prints value / per-layer ms
for v in "$@"; do
  KWS_AMD_LIB=$GRAFT_REPO_ROOT/variants/libkws_$v.so python bench.py $BENCH_ARGS --no-cpu-baseline --steps 10 --warmup 3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value']/1e6,1), [round(x,3) for x in d['roofline']['per_layer_ms']])"
done
