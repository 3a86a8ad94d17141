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
