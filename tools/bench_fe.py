#!/usr/bin/env python3
"""Front-end alone: B streams x one 3600+240-sample chunk -> mel; prints ms per call (HIP events).
usage: bench_fe.py [B] [samples] [buffers]   buffers > 1 rotates the input over that many PCM buffers: the same buffer again and
again is served by the 256 MB Infinity Cache, a streaming loop's fresh chunk comes from HBM."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from keyword_spotting_amd import get_config
from keyword_spotting_amd.frontend import MelFrontend
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
N = int(sys.argv[2]) if len(sys.argv) > 2 else 3840
fe = MelFrontend(get_config())
NB = int(sys.argv[3]) if len(sys.argv) > 3 else 1
pcms = [torch.randn(B, N, device="cuda") * 0.1 for _ in range(NB)]
pcm = pcms[0]
for i in range(5): mel = fe.forward(pcms[i % NB])
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for i in range(50): mel = fe.forward(pcms[i % NB])
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / 50
print("front-end B=%d samples=%d buffers=%d -> T=%d: %.4f ms per call, %.1f M frames/s" % (B, N, NB, mel.shape[1], ms, B * mel.shape[1] / ms / 1e3))
