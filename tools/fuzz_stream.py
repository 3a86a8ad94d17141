#!/usr/bin/env python3
"""Randomised soak of the native loop (kws_stream_feed) against the HotwordDetector mirror of detector.py:158-209: random
chunk lengths (empty, sub-frame, odd, long), int16 / float PCM, silent stretches, several window sizes and batch sizes,
fp32, f16x3 and bf16 stacks.  Every chunk: identical hits and bit-identical recurrent state.  usage: fuzz_stream.py [seeds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from keyword_spotting_amd import get_config, weights
from keyword_spotting_amd.detector import HotwordDetector, StreamManager
from keyword_spotting_amd.frontend import MelFrontend
from keyword_spotting_amd.rnn_ctc import DeployModel

seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
bad = 0
for seed in range(seeds):
    rng = np.random.default_rng(1000 + seed)
    prec = ("fp32", "f16x3", "bf16", "fp32", "f16x3")[seed % 5]
    n_mel = 60 if seed % 4 == 1 else 40
    cfg = get_config(precision=prec, n_mel=n_mel)
    w = weights.init_weights(cfg, seed=seed)
    w["Wfc"] = (w["Wfc"] * 3).astype(np.float32)
    fe = MelFrontend(cfg)
    b = int(rng.integers(1, 40))
    win = int(rng.choice([1, 3, 15, 33]))
    label = str(int(rng.integers(1, 5)))
    det = HotwordDetector(DeployModel(cfg, w), batch=b, label=label, window_chunks=win)
    mgr = StreamManager(DeployModel(cfg, w), b, label=label, window_chunks=win)
    as_int16 = bool(seed % 2)
    fired = 0
    for c in range(40):
        n = int(rng.choice([0, rng.integers(1, 160), rng.integers(160, 400), rng.integers(400, 5001), 3600, 3600]))
        x = rng.standard_normal((b, n)) * 0.2
        x[rng.random(b) < 0.2] *= 1e-4
        piece = torch.from_numpy((x * 32768).clip(-32768, 32767).astype(np.int16)) if as_int16 else torch.from_numpy(x.astype(np.float32))
        want = np.zeros(b, np.int32)
        want[det.feed_pcm(piece, fe)] = 1
        got = mgr.feed_pcm(piece, fe).cpu().numpy()
        if not (np.array_equal(got, want) and torch.equal(mgr.state, det.state)):
            print("MISMATCH seed %d chunk %d n=%d prec=%s int16=%s b=%d win=%d" % (seed, c, n, prec, as_int16, b, win))
            bad += 1
            break
        fired += int(want.sum())
    else:
        print("seed %2d %s n_mel=%d int16=%d b=%2d window=%2d label=%s: 40 chunks ok, %d triggers" % (seed, prec, n_mel, as_int16, b, win, label, fired), flush=True)
    mgr.close()
print("FAILED" if bad else "all %d seeds agree" % seeds)
sys.exit(1 if bad else 0)
