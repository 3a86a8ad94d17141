"""TEST INFRASTRUCTURE ONLY -- builds oracle/kws_oracle.c into oracle/_build/libkws_oracle.so and
wraps it with ctypes.  `python oracle/build.py` or oracle.build.load().

No oracle/_ref here: the only compiled reference source on this path,
octbit/octbit_mat_mul_op.cc, includes tensorflow/core/framework/op_kernel.h and registers a TF
OpKernel; TensorFlow headers are absent from the image, so it is UNBUILDABLE."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_OUT = os.path.join(_HERE, "_build")
_SO = os.path.join(_OUT, "libkws_oracle.so")
_SRC = os.path.join(_HERE, "kws_oracle.c")


def build(native=False, out=None, force=False):
    out = out or _SO
    os.makedirs(os.path.dirname(out), exist_ok=True)
    if not force and os.path.exists(out) and os.path.getmtime(out) >= os.path.getmtime(_SRC):
        return out
    cmd = ["gcc", "-O3", "-fPIC", "-shared", "-fopenmp", "-fno-math-errno",
           "-march=native" if native else "-msse4.2", _SRC, "-o", out, "-lm"]
    subprocess.check_call(cmd)
    return out


class OracleCfg(ctypes.Structure):
    _fields_ = [("n_mel", ctypes.c_int), ("hidden", ctypes.c_int), ("num_layers", ctypes.c_int),
                ("num_classes", ctypes.c_int), ("use_relu", ctypes.c_int),
                ("value_clip", ctypes.c_float)]


_f = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")


class Oracle(object):
    def __init__(self, path):
        self.lib = ctypes.CDLL(path)
        self.lib.oracle_gru_forward.restype = ctypes.c_int
        self.lib.oracle_ctc_decode2.restype = ctypes.c_int
        self.lib.oracle_octbit_matmul.restype = ctypes.c_int

    def gru_forward(self, cfg, blob, mel, state, seq_len=None, threads=1, want_softmax=True):
        mel = np.ascontiguousarray(mel, np.float32)
        state = np.ascontiguousarray(state, np.float32)
        b, t, _ = mel.shape
        c = OracleCfg(*cfg)
        logits = np.empty((b, t, c.num_classes), np.float32)
        sm = np.empty_like(logits) if want_softmax else None
        s_out = np.empty_like(state)
        sl = None if seq_len is None else np.ascontiguousarray(seq_len, np.int32)
        vp = lambda a: None if a is None else a.ctypes.data_as(ctypes.c_void_p)
        rc = self.lib.oracle_gru_forward(ctypes.byref(c), vp(blob), vp(mel), vp(state), vp(sl),
                                         vp(logits), vp(sm), vp(s_out), b, t, int(threads))
        assert rc == 0
        return logits, sm, s_out

    def ctc_decode2(self, softmax, classnum, thres=0.4):
        sm = np.ascontiguousarray(softmax, np.float32)
        out = np.zeros(max(sm.shape[0], 1), np.int32)
        n = self.lib.oracle_ctc_decode2(sm.ctypes.data_as(ctypes.c_void_p), sm.shape[0], classnum,
                                        ctypes.c_float(thres), out.ctypes.data_as(ctypes.c_void_p))
        return out[:n]

    def octbit_matmul(self, x, wq, scale, bias):
        x = np.ascontiguousarray(x, np.float32)
        wq = np.ascontiguousarray(wq, np.int8)
        bias = np.ascontiguousarray(bias, np.float32)
        out = np.empty((x.shape[0], wq.shape[0]), np.float32)
        rc = self.lib.oracle_octbit_matmul(x.ctypes.data_as(ctypes.c_void_p),
                                           wq.ctypes.data_as(ctypes.c_void_p), ctypes.c_float(scale),
                                           bias.ctypes.data_as(ctypes.c_void_p),
                                           out.ctypes.data_as(ctypes.c_void_p), x.shape[0],
                                           x.shape[1], wq.shape[0])
        return rc, out


def load(native=False, out=None):
    return Oracle(build(native=native, out=out))


if __name__ == "__main__":
    print(build(force=True))
