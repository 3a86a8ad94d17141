"""Generates tests/golden/decode_golden.npz by IMPORTING the reference's numpy-only modules
(utils/prediction.py, utils/queue.py, utils/basic_vad.py) from /root/reference.

Runs in the build container only (the reference tree never travels to the GPU box); the .npz it
writes holds inputs and the reference's outputs -- data, no reference source.

    python tests/golden/make_decode_golden.py
"""
import os
import sys

sys.dont_write_bytecode = True
os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.path.insert(0, "/root/reference")

import numpy as np  # noqa: E402
from utils.prediction import (ctc_decode, ctc_decode2, ctc_decode_strict,  # noqa: E402
                              ctc_predict, evaluate)
from utils.queue import SimpleQueue  # noqa: E402
from utils.basic_vad import vad  # noqa: E402

C = 6
out = {}
cases = []


def plateau(words, hold, gap=0, p=0.9, other=None):
    """softmax rows spelling `words` (1..4), each held `hold` frames, `gap` blank frames between."""
    rows = []
    for w in words:
        for _ in range(hold):
            r = np.full(C, (1.0 - p) / (C - 1), np.float32)
            r[w] = p
            if other is not None:
                r[other[0]] = other[1]
            rows.append(r)
        for _ in range(gap):
            r = np.full(C, 0.02, np.float32)
            r[C - 1] = 0.9
            rows.append(r)
    return np.asarray(rows, np.float32).reshape(-1, C)


rng = np.random.default_rng(20171)
# (i) seeded random, peaked and flat
for t in (0, 1, 2, 3, 22, 300, 337):
    for alpha in (0.1, 0.3, 1.0):
        if t == 0:
            cases.append(np.zeros((0, C), np.float32))
        else:
            cases.append(rng.dirichlet([alpha] * C, t).astype(np.float32))
# smoothed random walks (long plateaus -> lockout / repeated emission / loose mode)
for t in (60, 300):
    for _ in range(6):
        words = rng.integers(1, 5, size=t // 5 + 1)
        hold = rng.integers(1, 9)
        cases.append(plateau(words, int(hold), int(rng.integers(0, 4)),
                             p=float(rng.uniform(0.45, 0.95)))[:t])
# (ii) planted keyword patterns
cases.append(plateau([1, 2, 3, 3], 4))                 # decode2 -> 1,2,3,3 ; ctc_decode doubles
cases.append(plateau([1, 2, 3, 3], 3))
cases.append(plateau([1, 2, 3, 3], 2, gap=2))
cases.append(plateau([1, 2, 3, 3], 1, gap=3))
cases.append(plateau([1, 2, 4, 3, 3], 2, gap=1))       # garbage 4 interleaved: 12433 no match
cases.append(plateau([1, 2, 3], 1, gap=3))             # enters loose mode, ends there
cases.append(plateau([1, 2, 3, 1, 2, 3, 3], 1, gap=4))
cases.append(plateau([2, 1, 2, 3, 3, 4], 1, gap=2, p=0.7))
# loose-mode branches: after 1,2,3 feed low rows / le4 > 0.2 rows / strong other word
base = plateau([1, 2, 3], 1, gap=2)


def rows(*specs):
    r = []
    for s in specs:
        v = np.full(C, 0.01, np.float32)
        for k, val in s.items():
            v[k] = val
        r.append(v)
    return np.asarray(r, np.float32)


cases.append(np.concatenate([base, rows({3: 0.25}, {1: 0.9}, {1: 0.9})]))
cases.append(np.concatenate([base, rows({1: 0.1}, {1: 0.1}, {3: 0.21}, {2: 0.9})]))
cases.append(np.concatenate([base, rows({1: 0.65}, {1: 0.65}, {1: 0.65}, {1: 0.65}, {1: 0.65},
                                        {3: 0.3}, {2: 0.7})]))
cases.append(np.concatenate([base, rows({2: 0.61, 3: 0.19}, {2: 0.61, 3: 0.19}, {4: 0.7},
                                        {4: 0.7}, {4: 0.7}, {4: 0.7}, {4: 0.1}, {1: 0.9})]))
cases.append(np.concatenate([plateau([1, 2, 3, 3], 1, gap=2),
                             rows({1: 0.1}, {1: 0.1}, {1: 0.1}, {1: 0.1}, {2: 0.55}, {3: 0.19})]))
# (iii) threshold edges (exact float32 constants) and argmax ties
edge = np.zeros((10, C), np.float32)
edge[0, 1] = np.float32(0.4)                           # == 0.4 : not > thres for decode2
edge[1, 1] = np.nextafter(np.float32(0.4), np.float32(1))
edge[2, 2] = np.float32(0.5)
edge[3, 2] = np.nextafter(np.float32(0.5), np.float32(1))
edge[4, 1] = edge[4, 3] = np.float32(0.45)             # tie -> first max
edge[5, 4] = edge[5, 2] = np.float32(0.5000001)
edge[6, 0] = np.float32(0.99)                          # space column is ignored
edge[7, 5] = np.float32(0.99)                          # blank column is ignored
edge[8, 3] = np.float32(0.2)
edge[9, 3] = np.nextafter(np.float32(0.2), np.float32(1))
cases.append(edge)
cases.append(edge[::-1].copy())

out["n_cases"] = np.int64(len(cases))
for i, sm in enumerate(cases):
    sm = np.ascontiguousarray(sm, np.float32)
    out["c%d_softmax" % i] = sm
    d2 = ctc_decode2(sm, C)
    d1 = ctc_decode(sm)
    ds = ctc_decode_strict(sm, C)
    out["c%d_decode2" % i] = d2
    out["c%d_decode" % i] = d1
    out["c%d_strict" % i] = ds
    out["c%d_decode2_t03" % i] = ctc_decode2(sm, C, thres=0.3)
    out["c%d_decode_l5" % i] = ctc_decode(sm, lockout=5, thres=0.45, loose_thres=0.25)
    out["c%d_strict_l2" % i] = ctc_decode_strict(sm, C, lockout=2, thres=0.6)
    out["c%d_predict" % i] = np.asarray(
        [ctc_predict(d2), ctc_predict(d1), ctc_predict(ds), ctc_predict(d2, "12"),
         ctc_predict(d1, "33")], np.int32)

# ctc_predict on hand-made sequences (negative terminator, garbage kept)
pseqs = [[0, 1, 0, 2, 0, 3, 0, 3, 0], [0, 1, 0, 2, 0, 4, 0, 3, 0, 3, 0], [1, 2, -1, 3, 3],
         [0], [], [1, 2, 3, 3], [3, 1, 2, 3, 3, 1], [1, 2, 3, -1, 3], [0, 0, 1, 0, 2, 0, 3, 3]]
out["n_pseq"] = np.int64(len(pseqs))
for i, s in enumerate(pseqs):
    out["p%d_seq" % i] = np.asarray(s, np.int32)
    out["p%d_out" % i] = np.asarray([ctc_predict(s), ctc_predict(s, "123"), ctc_predict(s, "33")],
                                    np.int32)

# evaluate
res = rng.integers(0, 2, 64).tolist()
tgt = rng.integers(0, 2, 64).tolist()
out["eval_result"] = np.asarray(res, np.int32)
out["eval_target"] = np.asarray(tgt, np.int32)
out["eval_out"] = np.asarray(evaluate(res, tgt), np.int32)

# (iv) SimpleQueue trace: ops 0=add(k) 1=clear ; record (len, full, contents) after each op
ops = rng.choice([0, 0, 0, 0, 0, 1], size=80)
ops[:20] = 0
q = SimpleQueue(15)
trace_len, trace_full, trace_head, trace_n = [], [], [], []
for k, op in enumerate(ops):
    if op == 0:
        q.add(k)
    else:
        q.clear()
    trace_len.append(q.len)
    trace_full.append(int(q.full()))
    content = q.get_all()
    trace_n.append(len(content))
    trace_head.append(content[0] if content else -1)
out["q_ops"] = np.asarray(ops, np.int32)
out["q_len"] = np.asarray(trace_len, np.int32)
out["q_full"] = np.asarray(trace_full, np.int32)
out["q_n"] = np.asarray(trace_n, np.int32)
out["q_head"] = np.asarray(trace_head, np.int32)

# (v) vad around the call-site threshold 30 (detector.py:168) and default 40
sig = rng.standard_normal((24, 3600)).astype(np.float32)
scale = np.linspace(0.001, 0.02, 24).astype(np.float32)[:, None]
sig = sig * scale
sig[0] = 0.0
sig[1, :300] = 0.1
sig[1, 300:] = 0.0            # sum |x| == 30.000.. edge (float32 accumulation)
out["vad_sig"] = sig
out["vad_30"] = np.asarray([int(vad(s, 30)) for s in sig], np.int32)
out["vad_40"] = np.asarray([int(vad(s)) for s in sig], np.int32)
out["vad_sum"] = np.asarray([np.abs(s).sum() for s in sig], np.float32)

dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "decode_golden.npz")
np.savez_compressed(dst, **out)
print("wrote", dst, "cases:", len(cases), "bytes:", os.path.getsize(dst))
loose_hits = sum(1 for i in range(len(cases))
                 if not np.array_equal(out["c%d_decode" % i], out["c%d_strict" % i]))
print("cases where ctc_decode != ctc_decode_strict (loose mode exercised):", loose_hits)
print("predict hits:", sum(int(out["c%d_predict" % i][0]) for i in range(len(cases))))
