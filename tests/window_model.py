"""TEST INFRASTRUCTURE -- a plain-Python statement of the incremental decode window that csrc/window_device.h implements on
the device, so that its equivalence with the reference's windowed re-scan (detector.py:195-209: SimpleQueue.add, concatenate,
ctc_decode2, ctc_predict) can be property-tested on the CPU.  Not imported by the product.

Per queued chunk the window keeps a SUMMARY instead of the chunk's frames:
    n        frames in the chunk (0 = an empty chunk: it takes a slot, utils/queue.py:26-32, but changes nothing else)
    first    ctc_decode2 frame word of its first frame (-1 = nothing above the threshold)
    last     ... of its last frame
    tab[q]   where the label matcher ends up when it enters the chunk's frames 1..n-1 in state q (q = number of label digits
             matched so far, 0..len-1; len = the label has occurred, absorbing) -- the chunk's emissions AFTER its first frame
             do not depend on anything outside the chunk (utils/prediction.py:74-80: pre_word is the previous frame's word)
Whether the chunk's FIRST frame emits does depend on the chunk before it in the window (its word must differ from that
chunk's last word, or there is none) -- and that is exactly what changes when the oldest chunk is evicted, so it is decided
when the window is evaluated, not when the chunk is summarised.  Evaluation walks the <= maxLen summaries, oldest first:
O(chunks) per step instead of O(frames in the window).
"""


def kmp_delta(label_digits, n_words=9):
    """delta[q][w] for q in 0..len-1 and emitted word w in 1..n_words: matcher state after reading w in state q."""
    lab = list(label_digits)
    n = len(lab)
    delta = [[0] * (n_words + 1) for _ in range(max(n, 1))]
    for q in range(n):
        for w in range(1, n_words + 1):
            k = q + 1
            # longest k such that lab[:k] is a suffix of lab[:q] + [w]
            s = lab[:q] + [w]
            while k > 0 and s[len(s) - k:] != lab[:k]:
                k -= 1
            delta[q][w] = k
    return delta


def summarise(words, delta, n_label):
    """(n, first, last, tab) of one chunk's frame words."""
    n = len(words)
    if n == 0:
        return 0, -1, -1, list(range(max(n_label, 1)))
    tab = []
    for q0 in range(max(n_label, 1)):
        q, pre = q0, words[0]
        for w in words[1:]:
            if w >= 0 and w != pre and q < n_label:
                q = delta[q][w + 1]
            pre = w
        tab.append(q)
    return n, words[0], words[-1], tab


class IncrementalWindow(object):
    """The summary ring exactly as csrc/window_device.h keeps it.  Besides tab (the chunk as the OLDEST of the window: its first
    frame is decided at evaluation time) every slot carries ftab: the chunk's whole table with its first frame already decided
    against the last word of its predecessor -- the nearest earlier non-empty chunk at the time it was queued.  Evictions are
    FIFO, so that predecessor is either still queued (ftab is right) or gone together with everything before it (the chunk
    is then the first non-empty one and is evaluated through tab).  Evaluation = one table lookup per queued chunk."""

    def __init__(self, max_chunks, label_digits):
        self.nq, self.label = int(max_chunks), list(label_digits)
        self.delta = kmp_delta(self.label)
        self.ring = []                       # (n, first, last, tab, ftab), oldest first

    def step(self, words, clear_before=False):
        """One chunk: [clear]; add; evaluate; on a hit clear.  -> 1 / 0."""
        n_label = len(self.label)
        if clear_before:
            self.ring = []
        pred = [e for e in self.ring if e[0] > 0]
        pred_last = pred[-1][2] if pred else -1               # (may be the chunk evicted right below: ftab is then never used)
        if len(self.ring) == self.nq:
            self.ring.pop(0)
        n, first, last, tab = summarise(list(words), self.delta, n_label)
        states = max(n_label, 1)
        if n == 0:
            ftab = list(range(states))
        elif first >= 0 and first != pred_last:
            ftab = [tab[self.delta[q][first + 1]] if self.delta[q][first + 1] < n_label else n_label for q in range(states)]
        else:
            ftab = list(tab)
        self.ring.append((n, first, last, tab, ftab))
        if n_label == 0:
            hit = True                        # '' is a substring of anything (utils/prediction.py:118)
        else:
            q, seen = 0, False
            for n, first, last, tab, ftab in self.ring:
                if q == n_label:
                    break                     # absorbing
                if n == 0:
                    continue
                if not seen:                  # the first non-empty chunk: nothing before it in the window
                    seen = True
                    if first >= 0:
                        q = self.delta[q][first + 1]
                        if q == n_label:
                            break
                    q = tab[q]
                else:
                    q = ftab[q]
            hit = q == n_label
        if hit:
            self.ring = []
        return int(hit)
