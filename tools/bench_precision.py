"""ms per step and per layer kernel of one precision at the headline shape: python tools/bench_precision.py f16x3 [B] [T]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from keyword_spotting_amd import get_config, weights
from keyword_spotting_amd.rnn_ctc import DeployModel

prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
T = int(sys.argv[3]) if len(sys.argv) > 3 else 300
cfg = get_config(precision=prec)
m = DeployModel(cfg, weights.init_weights(cfg, seed=0))
mel = (torch.randn(B, T, cfg.n_mel, device="cuda").abs() * 2).contiguous()
st, pw = m.zero_state(B), m.fresh_prev_word(B)
m.reserve(B, T)
for _ in range(3):
    m.forward(mel, st, prev_word=pw, state_out=st)
torch.cuda.synchronize()
m.set_profiling(True)
m.kernel_times()
t0 = time.perf_counter()
n = 10
for _ in range(n):
    m.forward(mel, st, prev_word=pw, state_out=st)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
kt = m.kernel_times()
print("%s B=%d T=%d: %.3f ms/step = %.1f M frames/s; kernels %s : %s" % (
    prec, B, T, dt * 1e3, B * T / dt / 1e6, m.kernel_names(), ["%.3f ms" % (k[0] / max(k[1], 1)) for k in kt]), flush=True)
