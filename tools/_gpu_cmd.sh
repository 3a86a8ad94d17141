python -m pytest tests -m gpu -q -x -k "detector or frontend" 2>&1 | tail -6
python tools/bench_e2e.py
python tools/bench_e2e.py --precision bf16
