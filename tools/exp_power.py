#!/usr/bin/env python3
"""Is the fp32 kernel power/clock limited?  Same binary, same instruction stream: per-layer kernel time with random
weights + random mel vs all-zero weights + zero mel (data-dependent power; MI355X_MICROARCH.md 'DVFS give-back')."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from keyword_spotting_amd import get_config, weights
from keyword_spotting_amd.rnn_ctc import DeployModel
cfg = get_config()
B, T = 4096, 300
def run(tag, w, mel):
    m = DeployModel(cfg, w)
    st = m.zero_state(B)
    for _ in range(3): m.forward(mel, st, state_out=st)
    torch.cuda.synchronize()
    m.set_profiling(True); m.kernel_times(reset=True)
    for _ in range(10): m.forward(mel, st, state_out=st)
    kt = m.kernel_times()
    print("%-28s per-layer ms %s" % (tag, " ".join("%.4f" % (a / n) for a, n in kt)), flush=True)
    m.close()
w = weights.init_weights(cfg, seed=0)
mel = (torch.randn(B, T, 40, device="cuda").abs() * 2).contiguous()
wz = dict(layers=[{k: np.zeros_like(v) for k, v in l.items()} for l in w["layers"]], Wfc=np.zeros_like(w["Wfc"]), bfc=np.zeros_like(w["bfc"]))
for rep in range(2):
    run("random weights, random mel", w, mel)
    run("zero weights, zero mel", wz, torch.zeros_like(mel))
    run("random weights, zero mel", w, torch.zeros_like(mel))
