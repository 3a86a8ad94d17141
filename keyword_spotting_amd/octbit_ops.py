"""octbit_mat_mul -- octbit/octbit_ops.py:17-26 backed by the HIP kernel (kws_octbit_matmul)."""
import numpy as np
import torch

from . import _lib


def octbit_mat_mul(x1, x2, transpose_a=False, transpose_b=True, scale=0.0, bias=(0,), per_row_scale=False):
    """x1 float [A,K]; x2 int8 [N,K] (already transposed); scale > 0; bias [N].
    Error behaviour follows octbit/octbit_mat_mul_op.cc:41-46,61-73."""
    lib = _lib.load()
    if not transpose_b:
        raise _lib.InvalidArgumentError(-1, "b need to be transposed")
    if transpose_a:
        raise _lib.InvalidArgumentError(-1, "a cannot to be transposed")
    x = torch.as_tensor(np.asarray(x1, np.float32) if not torch.is_tensor(x1) else x1).to(torch.float32)
    w = torch.as_tensor(np.asarray(x2, np.int8) if not torch.is_tensor(x2) else x2).to(torch.int8)
    if x.dim() != 2:
        raise _lib.InvalidArgumentError(-1, "In[0] is not a matrix")
    if w.dim() != 2:
        raise _lib.InvalidArgumentError(-1, "In[1] is not a matrix")
    if x.shape[1] != w.shape[1]:
        raise _lib.InvalidArgumentError(-1, "f is not equal in filter and input")
    dev = x.device if x.is_cuda else torch.device("cuda:0")
    x, w = x.to(dev).contiguous(), w.to(dev).contiguous()
    b = torch.as_tensor(np.asarray(bias, np.float32) if not torch.is_tensor(bias) else bias).to(torch.float32).to(dev).contiguous()
    a, k, n = int(x.shape[0]), int(x.shape[1]), int(w.shape[0])
    if b.numel() != n:
        raise _lib.InvalidArgumentError(-1, "bias must have %d entries, got %d" % (n, b.numel()))
    out = torch.empty(a, n, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.kws_octbit_matmul(_lib.ptr(x), _lib.ptr(w), float(scale), _lib.ptr(b), _lib.ptr(out),
                                         a, k, n, int(bool(per_row_scale)), _lib.current_stream_ptr()))
    return out
