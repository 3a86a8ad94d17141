for b in 16 64 256 1024 2048 4096; do python bench.py --batch $b --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('B=%5d  value %.1f M  per-layer ms %s' % (d['config']['streams_per_gpu'], d['value']/1e6, d['roofline']['per_layer_ms']))"; done
python -m pytest tests/test_gpu_fullsize.py -q 2>&1 | tail -5
