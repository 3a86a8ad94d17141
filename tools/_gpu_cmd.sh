python -m pytest tests -m gpu -q -k "soak or parity or fullsize" 2>&1 | tail -3
for i in 1 2 3; do python tools/stress_determinism.py 2>&1 | tail -2; done
python tools/bench_config.py; python tools/bench_config.py
python tools/bench_config.py --n-mel 40 --hidden 128 --layers 2 --batch 2048 --kernel generic
python tools/bench_config.py --n-mel 40 --hidden 64 --layers 8 --batch 512
