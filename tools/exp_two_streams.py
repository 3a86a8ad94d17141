#!/usr/bin/env python3
"""Experiment: the PCM -> trigger loop for 4096 streams as ONE chain of kernels (StreamManager, B=4096) or as TWO chains of
2048 streams on two HIP streams (the small kernels and launch gaps of one chain under the GRU kernels of the other)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from keyword_spotting_amd import get_config, weights
from keyword_spotting_amd.detector import StreamManager
from keyword_spotting_amd.frontend import MelFrontend
from keyword_spotting_amd.rnn_ctc import DeployModel

prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
cfg = get_config(precision=prec)
w = weights.init_weights(cfg)
B = 4096

def run(parts, chunks=40):
    nb = B // parts
    mgrs = [StreamManager(DeployModel(cfg, w), nb) for _ in range(parts)]
    fes = [MelFrontend(cfg) for _ in range(parts)]
    streams = [torch.cuda.Stream() for _ in range(parts)]
    pcm = [[(torch.randn(nb, 3600, device="cuda") * 0.1).contiguous() for _ in range(4)] for _ in range(parts)]
    def one(c):
        for i in range(parts):
            with torch.cuda.stream(streams[i]):
                mgrs[i].feed_pcm(pcm[i][c % 4], fes[i])
    for c in range(5): one(c)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for c in range(chunks): one(c)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / chunks
    print("%s, %d chain(s) of %d streams: %.3f ms per 225 ms chunk of all %d streams -> %.2f M real-time streams" % (prec, parts, nb, dt * 1e3, B, B * 0.225 / dt / 1e6))
    for m in mgrs: m.close()

run(1); run(2); run(1); run(2); run(4)
