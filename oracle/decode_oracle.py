"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's greedy CTC collapse, keyword
decision, decode window, VAD and chunk/carry arithmetic.  Pure Python/numpy, small inputs only.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

PARITY PINNED: every function here is checked by tests/test_oracle_decode.py against golden
vectors produced by importing the reference's own numpy code in the build container
(tests/golden/make_decode_golden.py -> tests/golden/decode_golden.npz).

Each decoder is written as an explicit per-frame state machine (the form the HIP kernels use),
not as the reference's index-skipping while loop; equality with the reference is what the golden
vectors establish.
"""
import numpy as np


def _wrap(tokens):
    """Reference output format [0, w1, 0, w2, 0, ...] int32 (utils/prediction.py:58-62)."""
    out = np.zeros(2 * len(tokens) + 1, np.int32)
    out[1::2] = tokens
    return out


def frame_words(softmax, lo, hi, thres):
    """Per frame: index (0-based inside columns lo..hi-1) of the first maximum if that maximum is
    strictly above `thres`, else -1.  Shared first stage of all three decoders."""
    p = np.asarray(softmax)[:, lo:hi]
    if p.shape[0] == 0:
        return np.zeros(0, np.int64)
    best = p.argmax(axis=1)
    return np.where(p.max(axis=1) > thres, best, -1)


def ctc_decode2(softmax, classnum, thres=0.4):
    """utils/prediction.py:65-86 -- live streaming decoder.
    A frame emits word w+1 iff its word w != -1 and differs from the previous frame's word."""
    w = frame_words(softmax, 1, classnum - 1, thres)
    prev = np.concatenate([[-1], w[:-1]]) if len(w) else w
    return _wrap((w[(w >= 0) & (w != prev)] + 1).tolist())


def ctc_decode_strict(softmax, classnum, lockout=3, thres=0.5):
    """utils/prediction.py:89-108 -- emit on threshold crossing, then ignore `lockout` frames
    (the emitting frame included)."""
    w = frame_words(softmax, 1, classnum - 1, thres)
    toks, skip_until = [], 0
    for i, wi in enumerate(w):
        if i < skip_until or wi < 0:
            continue
        toks.append(int(wi) + 1)
        skip_until = i + lockout
    return _wrap(toks)


def ctc_decode(softmax, lockout=3, thres=0.5, loose_thres=0.2):
    """utils/prediction.py:18-62 -- validation decoder with lockout and the 'loose' mode entered
    after the emitted history ends in 1,2,3.  Columns are hard-coded 1:5 (:21)."""
    p = np.asarray(softmax)[:, 1:5]
    n = p.shape[0]
    toks, when = [], []
    loose = False
    skip_until = 0
    for i in range(n):
        if i < skip_until:
            continue
        row = p[i]
        top = row.max()
        if not loose:
            if top > thres:
                toks.append(int(row.argmax()) + 1)
                when.append(i)
                skip_until = i + lockout
                loose = toks[-3:] == [1, 2, 3]
            continue
        # loose mode
        if top < loose_thres:
            if toks[-1] != 3:
                skip_until = i + lockout
                loose = False
        elif row[2] > loose_thres:
            toks.append(3)
            when.append(i)
            skip_until = i + lockout
            loose = False
        else:
            k = int(row.argmax())
            if row[k] > 0.6 and when[-1] + lockout < i:
                toks.append(k + 1)
                when.append(i)
    return _wrap(toks)


def ctc_predict(seq, label="1233"):
    """utils/prediction.py:111-118 -- digits of the positive entries (stop at the first negative)
    joined into a string; 1 iff `label` occurs in it."""
    digits = []
    for v in seq:
        if v < 0:
            break
        if v > 0:
            digits.append(str(int(v)))
    return 1 if label in "".join(digits) else 0


def evaluate(result, target):
    """utils/prediction.py:203-210 -- (misses, positives, false accepts) over 0/1 lists."""
    assert len(result) == len(target)
    miss = sum(1 for r, t in zip(result, target) if t and not r)
    fa = sum(1 for r, t in zip(result, target) if r and not t)
    return miss, sum(target), fa


class SimpleQueue(object):
    """utils/queue.py:16-38 -- bounded FIFO, drop-oldest when full.  `len` counts up to maxLen
    and is never decremented by eviction."""

    def __init__(self, maxLen):
        self.maxLen = maxLen
        self.clear()

    def clear(self):
        self.content, self.len = [], 0

    def full(self):
        return self.len == self.maxLen

    def add(self, item):
        if self.full():
            self.content.pop(0)
        else:
            self.len += 1
        self.content.append(item)

    def get_all(self):
        return self.content


def vad(sig, thres=40):
    """utils/basic_vad.py:17-18 -- sum of absolute sample values above threshold."""
    return bool(np.abs(np.asarray(sig)).sum() > thres)


def carry_len(n_samples, fft_size=400, hop_size=160):
    """detector.py:181-183 -- number of trailing samples carried into the next chunk."""
    return (n_samples - fft_size) % hop_size + (fft_size - hop_size)


def frames_in(n_samples, fft_size=400, hop_size=160):
    """utils/stft.py:27-81 framing (no padding): frames a buffer of n samples yields."""
    return 0 if n_samples < fft_size else (n_samples - fft_size) // hop_size + 1


def chunk_frame_counts(chunk_sizes, fft_size=400, hop_size=160):
    """detector.py:179-183 replayed on sample counts only: frames produced per sess.run call."""
    carry, out = 0, []
    for n in chunk_sizes:
        total = carry + n
        out.append(frames_in(total, fft_size, hop_size))
        carry = carry_len(total, fft_size, hop_size)
    return out
