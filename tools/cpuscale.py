import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from oracle import build as obuild, gru_oracle as G
orc = obuild.load(native=True, out="/tmp/libo.so")
w = G.init_weights(); blob = G.weights_to_blob(w)
c = (40,128,2,6,0,-1.0)
for thr in (1, 8, 32, 64, 128, 256):
    n = thr * 8
    mel = G.synthetic_mel(n, 300, 40, seed=2); st = np.zeros((2, n, 128), np.float32)
    orc.gru_forward(c, blob, mel, st, threads=thr)
    t0 = time.perf_counter(); k = 0
    while time.perf_counter() - t0 < 2.0:
        orc.gru_forward(c, blob, mel, st, threads=thr); k += 1
    dt = time.perf_counter() - t0
    print(thr, "threads:", round(k * n * 300 / dt), "frames/s", "per-thread", round(k * n * 300 / dt / thr))
