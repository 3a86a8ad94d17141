"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the reference's int8 weight quantiser and of the
rule that picks which matmuls are quantised.  Only tests/ (and smoke/bench checkers) import this.

PINNED by the structural identity the reference's op relies on (bias[j] == 127 * sum_i Wq[i, j],
octbit/octbit_graph.py:202-204) and, end to end with oracle_octbit_matmul, by the known-answer
tests of octbit/octbit_ops_test.py:24-34,41-53 (tests/test_oracle_octbit.py).
octbit/octbit_graph.py itself cannot be imported here (it imports tensorflow at module scope).
"""
import numpy as np


def octize_weight_int8_signed(w):
    """octbit/octbit_graph.py:191-215.  w: float [K, N] (TF MatMul kernel layout).
    Returns (Wq int8 [N, K] -- transposed as the op wants it, scale float, bias float64 [N])."""
    w = np.asarray(w)
    nmax = max(abs(w.max()), abs(w.min()))           # :196
    # :197 -- under the reference's NumPy 1.x, float32-scalar / python-float is a float64 ...
    scale = float(nmax) / 127.0
    # :201 -- ... and float32-array / float64-scalar is evaluated in float32 (numpy round: half to even)
    q = np.round(np.asarray(w, np.float32) / np.float32(scale))
    bias = (q * 127).sum(axis=0).astype(float)       # :202-204
    return np.ascontiguousarray(q.T).astype(np.int8), float(scale), bias


def default_octbit_matmul_name_check(name):
    """octbit/octbit_graph.py:218-225: quantise a node iff it is a MatMul, is not the named
    softmax projection, and does not belong to rnn cell_0."""
    return name != "model/linear/linear/MatMul" and "MatMul" in name and "cell_0" not in name


def octbit_matmul_ref(x, wq, scale_w, bias):
    """octbit/octbit_mat_mul_op.cc:90-181 in numpy (loops; small shapes only)."""
    x = np.asarray(x, np.float32)
    a_rows, k = x.shape
    n = wq.shape[0]
    assert scale_w > 0 and k % 64 == 0
    mn, mx = np.float32(x.min()), np.float32(x.max())
    signed = bool(mn < 0)
    if signed:
        bscale = np.float32(max(-mn, mx)) / np.float32(127)
        # C round(): half away from zero, computed in double on the float quotient
        quo = (x / bscale).astype(np.float64)
        q = (np.sign(quo) * np.floor(np.abs(quo) + 0.5) + 127).astype(np.int64)
    else:
        bscale = np.float32(mx) / np.float32(254)
        quo = (x / bscale).astype(np.float64)
        q = (np.sign(quo) * np.floor(np.abs(quo) + 0.5)).astype(np.int64)
    q = q.astype(np.uint8).astype(np.int64)
    scale = np.float32(scale_w) * np.float32(bscale)
    out = np.zeros((a_rows, n), np.float32)
    w = wq.astype(np.int64)
    for j in range(n):
        for a in range(a_rows):
            pair = np.clip(q[a, 0::2] * w[j, 0::2] + q[a, 1::2] * w[j, 1::2], -32768, 32767)
            lanes = [int(pair[m::4].sum()) for m in range(4)]
            o = np.float32(0)
            for m in range(4):
                o = np.float32(o + np.float32(lanes[m]))
            if signed:
                o = np.float32(o - np.float32(bias[j]))
            out[a, j] = np.float32(o * scale)
    return out


def octbit_rows(x, wq, scale_w, bias, groups=None, saturate=True):
    """octbit/octbit_mat_mul_op.cc:90-181 vectorised over R independent calls.

    x [R,K]; rows sharing a value in `groups` ([R] ints) form ONE op call (its `[A,K]` input: one
    min/max over all of them, :92-99); groups=None makes every row its own call (the deployed graph runs
    batch 1, so each GRU matmul sees A == 1).  The four i32 SSE lanes (:141-175) add up exactly in float
    (|sum| < 2^24), so the lane fold is not modelled separately here -- octbit_matmul_ref keeps it.
    An all-zero call has bscale == 0 and divides 0/0 in the reference; its output is defined here as 0
    (what x86 yields: the NaN casts to q == 0, and the output scale is 0).
    saturate=False drops _mm_maddubs_epi16's int16 clamp of the pair sums: NOT the reference's arithmetic, only the
    what-if of tests/test_oracle_octbit.py (how much of the int8 error is the clamp, how much the 8 bits)."""
    x = np.ascontiguousarray(x, np.float32)
    r_rows, k = x.shape
    n = wq.shape[0]
    assert scale_w > 0 and k % 64 == 0
    groups = np.arange(r_rows) if groups is None else np.asarray(groups)
    uniq, inv = np.unique(groups, return_inverse=True)
    mn = np.full(len(uniq), np.inf, np.float32)
    mx = np.full(len(uniq), -np.inf, np.float32)
    np.minimum.at(mn, inv, x.min(axis=1))
    np.maximum.at(mx, inv, x.max(axis=1))
    signed_g = mn < 0
    bscale_g = np.where(signed_g, np.maximum(-mn, mx) / np.float32(127), mx / np.float32(254)).astype(np.float32)
    signed, bscale = signed_g[inv], bscale_g[inv]
    with np.errstate(divide="ignore", invalid="ignore"):
        quo = (x / bscale[:, None]).astype(np.float64)                  # float quotient, widened for round()
    quo = np.where(bscale[:, None] == 0, 0.0, quo)
    q = np.sign(quo) * np.floor(np.abs(quo) + 0.5) + np.where(signed, 127.0, 0.0)[:, None]
    q = q.astype(np.int64).astype(np.uint8).astype(np.int32)
    wt = np.ascontiguousarray(wq.T).astype(np.int32)                    # [K, N]
    pair = q[:, 0::2, None] * wt[None, 0::2, :] + q[:, 1::2, None] * wt[None, 1::2, :]
    acc = (np.clip(pair, -32768, 32767) if saturate else pair).sum(axis=1)     # [R, N] exact
    o = acc.astype(np.float32) - np.where(signed[:, None], np.asarray(bias, np.float32)[None, :], np.float32(0))
    scale = (np.float32(scale_w) * bscale).astype(np.float32)
    return (o.astype(np.float32) * scale[:, None]).astype(np.float32)
