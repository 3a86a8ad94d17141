#!/usr/bin/env python3
"""Per-layer kernel time (HIP events around each launch) against the frames per call: what a launch costs before its
first frame and per frame after it.  usage: bench_tsweep.py [B] [fp32|f16x3]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from keyword_spotting_amd import get_config, weights
from keyword_spotting_amd.rnn_ctc import DeployModel
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg = get_config(precision=sys.argv[2] if len(sys.argv) > 2 else "fp32")
model = DeployModel(cfg, weights.init_weights(cfg))
model.set_profiling(True)
for T in (1, 2, 4, 8, 16, 22, 23, 44, 100, 300):
    mel = torch.rand(B, T, cfg.n_mel, device="cuda")
    st = model.zero_state(B)
    for _ in range(3): model.forward(mel, st, want_logits=False, state_out=st)
    model.kernel_times()
    for _ in range(20): model.forward(mel, st, want_logits=False, state_out=st)
    kt = model.kernel_times()
    print("T=%3d  layer0 %.4f ms  layer1 %.4f ms  per frame %.2f + %.2f us" % (T, kt[0][0] / kt[0][1], kt[1][0] / kt[1][1],
          kt[0][0] / kt[0][1] / T * 1e3, kt[1][0] / kt[1][1] / T * 1e3))
