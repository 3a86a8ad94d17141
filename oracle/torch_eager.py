"""TEST INFRASTRUCTURE ONLY -- op-by-op torch-CPU formulation of the same GRU path.

Third independent restatement (cross-checked against oracle/gru_oracle.py) and the "TF-CPU
stand-in" timed by bench.py's cpu_baseline leg: like the reference's TF graph at batch 1 it
dispatches a framework op per matmul / activation / elementwise step of every 10 ms frame
(models/rnn_ctc.py:238-243 runs a TF while_loop over GRUCell ops).  PARITY UNPINNED (see
oracle/gru_oracle.py).
"""
import torch


def to_torch(w, dtype=torch.float32):
    t = lambda a: torch.from_numpy(a).to(dtype)
    return dict(layers=[{k: t(v) for k, v in lay.items()} for lay in w["layers"]],
                Wfc=t(w["Wfc"]), bfc=t(w["bfc"]))


@torch.no_grad()
def gru_forward(tw, mel, state):
    """mel [B,T,I] tensor, state [L,B,H] tensor -> (logits [B,T,C], softmax, state')."""
    hs = list(torch.unbind(state, 0))
    hdim = hs[0].shape[1]
    tops = []
    for t in range(mel.shape[1]):
        x = mel[:, t, :]
        for l, lay in enumerate(tw["layers"]):
            h = hs[l]
            g = torch.sigmoid(torch.addmm(lay["bg"], torch.cat([x, h], 1), lay["Wg"]))
            r, u = torch.split(g, hdim, dim=1)
            c = torch.tanh(torch.addmm(lay["bc"], torch.cat([x, r * h], 1), lay["Wc"]))
            h = u * h + (1 - u) * c
            hs[l] = h
            x = h
        tops.append(x)
    top = torch.stack(tops, 1)
    logits = (top.reshape(-1, hdim) @ tw["Wfc"] + tw["bfc"]).reshape(mel.shape[0], mel.shape[1], -1)
    return logits, torch.softmax(logits, -1), torch.stack(hs)


def eager_stream_rate(args):
    """One pinned worker of bench.py's cpu_baseline leg (the eager stand-in on N cores, one
    independent batch-1 stream per core): the detector-style loop -- 22-frame chunks, state round-tripping through
    numpy -- for `seconds`, on CPU `cpu` (None: unpinned).  Returns mel-frames/s of this worker.  Top-level so that a
    spawn-context process pool can import it."""
    import os
    import time
    import numpy as np
    cpu, seconds, n_mel, hidden, layers, classes = args
    if cpu is not None:
        try:
            os.sched_setaffinity(0, {cpu})
        except OSError:
            pass
    torch.set_num_threads(1)
    from oracle import gru_oracle as G
    tw = to_torch(G.init_weights(n_mel, hidden, layers, classes, seed=0))
    mel = torch.from_numpy(G.synthetic_mel(1, 300, n_mel, seed=1))
    state = torch.zeros(layers, 1, hidden)
    for pos in range(0, 66, 22):                      # warm-up
        gru_forward(tw, mel[:, pos:pos + 22], state)
    frames, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for pos in range(0, 300, 22):
            lg, sm, state = gru_forward(tw, mel[:, pos:pos + 22], state)
            state = torch.from_numpy(np.array(state.numpy()))
            frames += lg.shape[1]
    return frames / (time.perf_counter() - t0)
