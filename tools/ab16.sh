#!/bin/bash
for v in "$@"; do
  KWS_AMD_LIB=$GRAFT_REPO_ROOT/variants/libkws_$v.so python bench.py --precision bf16 --no-cpu-baseline --steps 20 --warmup 5 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value']/1e6,1), [round(x,3) for x in d['roofline']['per_layer_ms']])"
done
