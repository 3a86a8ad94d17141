"""Offline int8 weight quantiser and the which-matmuls rule of octbit/octbit_graph.py:191-225
(the GraphDef rewriting around them is TF plumbing and out of scope)."""
import ctypes

import numpy as np

from . import _lib


def octize_weight_int8_signed(w):
    """octbit/octbit_graph.py:191-215 on a float [K,N] MatMul kernel.
    -> (Wq int8 [N,K] pre-transposed, scale float, bias float32 [N] = 127 * column sums of Wq)."""
    lib = _lib.load()
    w = np.ascontiguousarray(w, np.float32)
    if w.ndim != 2:
        raise _lib.InvalidArgumentError(-1, "weight must be a matrix")
    k, n = w.shape
    wq = np.empty((n, k), np.int8)
    bias = np.empty(n, np.float32)
    scale = ctypes.c_float()
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    _lib.check(lib.kws_octbit_quantize(vp(w), k, n, vp(wq), ctypes.byref(scale), vp(bias)))
    return wq, float(scale.value), bias


def default_octbit_matmul_name_check(name):
    """octbit/octbit_graph.py:218-225."""
    return name != "model/linear/linear/MatMul" and "MatMul" in name and "cell_0" not in name
