"""Greedy CTC collapse and keyword decision on the GPU -- same names and argument meaning as
utils/prediction.py.  Inputs are one stream's softmax [T,C] (the reference's form; the result is
then the reference's int32 [0,w1,0,w2,0,...] array) or a batch [B,T,C] (-> list of such arrays, or
raw device tensors with `raw=True`).  Everything runs in kws_ctc_decode / kws_ctc_predict."""
import numpy as np
import torch

from . import _lib


def _as_batch(softmax, device):
    sm = torch.as_tensor(softmax)
    if sm.dtype != torch.float32:
        sm = sm.to(torch.float32)
    single = sm.dim() == 2
    if single:
        sm = sm.unsqueeze(0)
    if sm.dim() != 3:
        raise _lib.InvalidArgumentError(-1, "softmax must be [T,C] or [B,T,C], got %s" % (tuple(sm.shape),))
    if not sm.is_cuda:
        sm = sm.to(device or "cuda:0")
    return sm.contiguous(), single


def decode_batch(kind, softmax, lengths=None, lockout=3, thres=0.5, loose_thres=0.2, max_words=None,
                 device=None):
    """-> (words [B,max_words] int32, counts [B] int32) device tensors."""
    lib = _lib.load()
    sm, _ = _as_batch(softmax, device)
    b, t, c = (int(v) for v in sm.shape)
    if max_words is None:
        max_words = max(t, 1)
    words = torch.zeros(b, max_words, dtype=torch.int32, device=sm.device)
    counts = torch.zeros(b, dtype=torch.int32, device=sm.device)
    if lengths is not None:
        lengths = torch.as_tensor(lengths).to(device=sm.device, dtype=torch.int32).contiguous()
    with torch.cuda.device(sm.device):
        _lib.check(lib.kws_ctc_decode(kind, _lib.ptr(sm), _lib.ptr(lengths), b, t, c, int(lockout),
                                      float(thres), float(loose_thres), _lib.ptr(words), _lib.ptr(counts),
                                      int(max_words), _lib.current_stream_ptr()))
    return words, counts


def _format(words, counts, single):
    w = words.cpu().numpy()
    n = counts.cpu().numpy()
    outs = []
    for row, k in zip(w, n):
        k = min(int(k), w.shape[1])
        seq = np.zeros(2 * k + 1, np.int32)          # utils/prediction.py:58-62
        seq[1::2] = row[:k]
        outs.append(seq)
    return outs[0] if single else outs


def ctc_decode(softmax, lockout=3, thres=0.5, loose_thres=0.2, raw=False):
    """utils/prediction.py:18."""
    words, counts = decode_batch(_lib.DECODE, softmax, None, lockout, thres, loose_thres)
    return (words, counts) if raw else _format(words, counts, torch.as_tensor(softmax).dim() == 2)


def ctc_decode2(softmax, classnum, thres=0.4, raw=False):
    """utils/prediction.py:65."""
    _check_classnum(softmax, classnum)
    words, counts = decode_batch(_lib.DECODE2, softmax, None, 3, thres, 0.0)
    return (words, counts) if raw else _format(words, counts, torch.as_tensor(softmax).dim() == 2)


def ctc_decode_strict(softmax, classnum, lockout=3, thres=0.5, raw=False):
    """utils/prediction.py:89."""
    _check_classnum(softmax, classnum)
    words, counts = decode_batch(_lib.DECODE_STRICT, softmax, None, lockout, thres, 0.0)
    return (words, counts) if raw else _format(words, counts, torch.as_tensor(softmax).dim() == 2)


def _check_classnum(softmax, classnum):
    c = int(torch.as_tensor(softmax).shape[-1])
    if int(classnum) != c:
        raise _lib.InvalidArgumentError(-1, "classnum=%d but softmax has %d columns" % (classnum, c))


def ctc_predict(seq, label="1233"):
    """utils/prediction.py:111 -- seq is a decoded [0,w,0,...] array (or any int sequence; a
    negative entry terminates it).  Returns 0/1.  A (words, counts) pair of device tensors returns a
    device tensor of hits."""
    lib = _lib.load()
    if isinstance(seq, tuple):
        words, counts = seq
        b, mw = int(words.shape[0]), int(words.shape[1])
        hit = torch.zeros(b, dtype=torch.int32, device=words.device)
        with torch.cuda.device(words.device):
            _lib.check(lib.kws_ctc_predict(_lib.ptr(words), _lib.ptr(counts), b, mw, label.encode(),
                                           _lib.ptr(hit), _lib.current_stream_ptr()))
        return hit
    vals = []
    for v in np.asarray(seq).ravel().tolist():
        if v < 0:
            break
        if v > 0:
            vals.append(int(v))
    if any(v > 9 for v in vals):
        raise _lib.InvalidArgumentError(-1, "ctc_predict words must be single digits")
    words = torch.tensor([vals + [0]], dtype=torch.int32, device="cuda:0")
    counts = torch.tensor([len(vals)], dtype=torch.int32, device="cuda:0")
    return int(ctc_predict((words, counts), label)[0].item())


def evaluate(result, target):
    """utils/prediction.py:203 -- (miss, positives, false_accept)."""
    if len(result) != len(target):
        raise AssertionError("result and target differ in length")
    r = np.asarray(result, bool)
    t = np.asarray(target, bool)
    return int((t & ~r).sum()), int(t.sum()), int((r & ~t).sum())


def tokens_to_seq(tokens_row):
    """Fused kws_step token events of one stream ([T] int8) -> the reference's [0,w,0,...] form."""
    row = np.asarray(torch.as_tensor(tokens_row).cpu())
    w = row[row > 0].astype(np.int32)
    seq = np.zeros(2 * len(w) + 1, np.int32)
    seq[1::2] = w
    return seq
