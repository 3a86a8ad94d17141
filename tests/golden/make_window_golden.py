"""Generates tests/golden/window_golden.npz: the decode-window part of the reference's loop body (detector.py:168-177 and
:195-209) driven with the reference's OWN functions -- utils/queue.py SimpleQueue, utils/prediction.py ctc_decode2 and
ctc_predict, imported from /root/reference -- over sequences of softmax chunks, with the per-chunk decision recorded.
detector.py itself is not importable (pyaudio, librosa, TensorFlow), so its six window lines are replayed here verbatim in
meaning:  [silence -> prob_queue.clear()];  prob_queue.add(softmax);  concatenate(get_all());  ctc_decode2(window, C);
ctc_predict(result, label);  on a hit: prob_queue.clear().

Runs in the build container only; the .npz holds inputs (per-frame softmax rows, chunk lengths, silence flags) and the
reference's outputs (hit per chunk, the decoded window sequence per chunk) -- data, no reference source.

    python tests/golden/make_window_golden.py
"""
import os
import sys

sys.dont_write_bytecode = True
os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.path.insert(0, "/root/reference")

import numpy as np  # noqa: E402
from utils.prediction import ctc_decode2, ctc_predict  # noqa: E402
from utils.queue import SimpleQueue  # noqa: E402

C = 6
rng = np.random.default_rng(20175)
out = {}


def rows_for(words, p=0.9):
    """softmax rows whose ctc_decode2 frame word is words[i] (0..3, or -1 = nothing above 0.4)."""
    r = np.full((len(words), C), 0.02, np.float32)
    for i, w in enumerate(words):
        if w < 0:
            r[i, C - 1] = 0.9                  # blank: ignored by the decoder
        else:
            r[i, w + 1] = p
    return r


def scenario(name, chunks, silent, maxlen, label):
    """chunks: list of word lists; silent: list of 0/1 (clear before the chunk is added)."""
    q = SimpleQueue(maxlen)
    hits, seqs = [], []
    for words, sil in zip(chunks, silent):
        if sil:
            q.clear()                                            # detector.py:171-177
        q.add(rows_for(words))                                   # :195
        window = np.concatenate(q.get_all(), 0)                  # :197
        result = ctc_decode2(window, C)                          # :200
        hit = int(ctc_predict(result, label))                    # :201
        hits.append(hit)
        seqs.append(np.asarray(result, np.int32))
        if hit:
            q.clear()                                            # :203
    out[name + "_words"] = np.asarray([w for c in chunks for w in c], np.int8)
    out[name + "_lens"] = np.asarray([len(c) for c in chunks], np.int32)
    out[name + "_silent"] = np.asarray(silent, np.uint8)
    out[name + "_maxlen"] = np.int64(maxlen)
    out[name + "_label"] = np.asarray([int(ch) for ch in label], np.int32)
    out[name + "_hit"] = np.asarray(hits, np.int32)
    out[name + "_seq_lens"] = np.asarray([len(s) for s in seqs], np.int32)
    out[name + "_seq"] = np.concatenate(seqs) if seqs else np.zeros(0, np.int32)


def held(words, hold_lo, hold_hi, gap_hi):
    """one long frame sequence: each word held a random number of frames, random gaps of -1 between."""
    seq = []
    for w in words:
        seq += [int(w)] * int(rng.integers(hold_lo, hold_hi + 1))
        seq += [-1] * int(rng.integers(0, gap_hi + 1))
    return seq


def cut(seq, lens):
    chunks, pos = [], 0
    for n in lens:
        chunks.append(seq[pos:pos + n])
        pos += n
    return [c for c in chunks]


names = []
# words that straddle chunk boundaries and evictions: long holds, chunk lengths 0..23, windows 1, 2, 3, 15
for k, (maxlen, label) in enumerate([(15, "1233"), (15, "1233"), (3, "1233"), (2, "12"), (1, "3"), (3, "11"), (15, "121"),
                                     (2, "1233"), (4, "33"), (15, "1233")]):
    words = rng.integers(0, 4, 120)
    if k in (0, 1, 9):                              # plant the keyword several times
        for pos in rng.integers(0, 110, 6):
            words[pos:pos + 4] = [0, 1, 2, 2]
    seq = held(words, 1, 9 if k % 2 else 30, 3 if k % 3 else 0)
    lens = []
    while sum(lens) < len(seq):
        lens.append(int(rng.choice([0, 1, 2, 5, 21, 22, 23, 23, 22])))
    chunks = cut(seq, lens)
    silent = (rng.random(len(chunks)) < (0.08 if k != 5 else 0.3)).astype(int).tolist()
    name = "w%d" % k
    scenario(name, chunks, silent, maxlen, label)
    names.append(name)
# the spelled keyword split over chunk boundaries exactly at the word changes, window full so that the first chunk is
# evicted on the step that would complete the keyword
chunks = [[0, 0], [0, 1], [1, 1], [2, 2], [-1, 2], [2, -1]]
scenario("w_split", chunks, [0] * len(chunks), 15, "1233")
names.append("w_split")
chunks = [[0] * 5] + [[1] * 3] + [[-1] * 2] * 13 + [[2, 2]] + [[-1, 2]]          # '1' falls out of a 15-chunk window before '33' arrives
scenario("w_evict", chunks, [0] * len(chunks), 15, "1233")
names.append("w_evict")
chunks = [[0, 0, 0]] + [[0] * 4] * 3 + [[1, 1]] + [[2]] + [[-1]] + [[2]]          # a word held across an evicted chunk is re-emitted by the re-scan
scenario("w_reemit", chunks, [0] * len(chunks), 2, "1233")
names.append("w_reemit")
chunks = [[0, 1], [], [], [2], [], [-1, 2], []]                                  # empty chunks take window slots (sub-frame PCM chunks)
scenario("w_empty", chunks, [0, 0, 1, 0, 0, 0, 0], 3, "1233")
names.append("w_empty")
out["names"] = np.asarray(names)
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "window_golden.npz")
np.savez_compressed(path, **out)
print("wrote", path, len(names), "scenarios,", sum(int(out[n + "_hit"].sum()) for n in names), "hits")
