python -m pytest tests -m gpu -q -k "soak or parity or fullsize" 2>&1 | tail -3
python tools/bench_config.py; python tools/bench_config.py
KWS_AMD_LIB=variants/libkws_vfb16.so python tools/bench_config.py
python tools/bench_config.py --batch 4096
KWS_AMD_LIB=variants/libkws_vfb16.so python tools/bench_config.py --batch 4096
python tools/bench_config.py --n-mel 40 --hidden 128 --layers 2 --batch 4096 --kernel generic
KWS_AMD_LIB=variants/libkws_vfb16.so python tools/bench_config.py --n-mel 40 --hidden 128 --layers 2 --batch 4096 --kernel generic
