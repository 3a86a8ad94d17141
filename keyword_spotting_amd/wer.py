"""Validation metric of the reference's `main.py` valid loop: label error rate by edit distance
(utils/wer.py:4-43 `wer`, :80-106 `WERCalculator`; used at main.py:43,201).  Host-side numpy: it runs once per
validation batch on a handful of short label rows, not on the streaming path.

Differences kept out on purpose: the reference's DP table is uint8 and silently wraps for sequences
longer than 254 labels (its docstring states the limit); here the distance is exact for any length and
equal to the reference's inside that limit."""
import numpy as np


def edit_distance(r, h):
    """Levenshtein distance between two label sequences (substitution = insertion = deletion = 1)."""
    r = np.asarray(r).reshape(-1)
    h = np.asarray(h).reshape(-1)
    prev = np.arange(len(h) + 1, dtype=np.int64)
    for i in range(1, len(r) + 1):
        # sub/del candidates are vectorised over the row; the insertion chain is a running minimum
        cand = np.minimum(prev[:-1] + (h != r[i - 1]), prev[1:] + 1)
        cur = np.empty_like(prev)
        cur[0] = i
        offs = np.arange(1, len(h) + 1, dtype=np.int64)
        # cur[j] = min(cand[j-1], cur[j-1] + 1)  ==  offs[j] + min-prefix(cand[k] - offs[k], i - 0)
        cur[1:] = np.minimum.accumulate(np.concatenate(([i], cand - offs)))[1:] + offs
        prev = cur
    return int(prev[-1])


def wer(r, h):
    """utils/wer.py:4-43: distance / len(r); the bare distance when the target is empty."""
    d = edit_distance(r, h)
    return float(d) if len(r) == 0 else float(d) / float(len(r))


class WERCalculator(object):
    """utils/wer.py:80-122.  `ignore_label_list` labels are skipped; a -1 ends the row."""

    def __init__(self, ignore_label_list):
        self._ignore = set(int(i) for i in ignore_label_list)

    def remove_residual(self, inputs):
        row = np.asarray(inputs).reshape(-1)
        end = np.flatnonzero(row == -1)
        if end.size:
            row = row[:end[0]]
        if self._ignore:
            row = row[~np.isin(row, sorted(self._ignore))]
        return row

    def cal_batch_wer(self, batch_r, batch_h):
        out = np.zeros(len(batch_r), np.float64)
        for i in range(len(batch_r)):
            r = self.remove_residual(batch_r[i])
            if len(r):
                out[i] = wer(r, self.remove_residual(batch_h[i]))
        return out

    def cal_topk_wers(self, batch_r, batch_h, batch_size, nums_gpu, topk, max_topk):
        """Best-of-top-k per utterance, hypotheses laid out [gpu][k][batch] (utils/wer.py:107-122)."""
        best = []
        for g in range(nums_gpu):
            r = batch_r[g * batch_size:(g + 1) * batch_size]
            h = batch_h[g * batch_size * max_topk:(g + 1) * batch_size * max_topk]
            per_k = [self.cal_batch_wer(r, h[k * batch_size:(k + 1) * batch_size]) for k in range(topk)]
            best.extend(np.min(np.vstack(per_k), axis=0))
        return best
