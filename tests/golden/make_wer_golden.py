"""Generates tests/golden/wer_golden.npz by IMPORTING the reference's utils/wer.py from /root/reference
(build container only).  The .npz holds label rows and the reference's outputs -- data, no source.

    python tests/golden/make_wer_golden.py
"""
import os
import sys

sys.dont_write_bytecode = True
os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.path.insert(0, "/root/reference")

import numpy as np  # noqa: E402
from utils.wer import WERCalculator, wer  # noqa: E402

rng = np.random.default_rng(4107)
out = {}
# (i) raw wer() on random label pairs, incl. empty target / empty hypothesis / equal rows
pairs_r, pairs_h, vals = [], [], []
for n in range(60):
    lr, lh = int(rng.integers(0, 12)), int(rng.integers(0, 12))
    r = rng.integers(1, 5, lr)
    h = r.copy() if n % 7 == 0 else rng.integers(1, 5, lh)
    pairs_r.append(np.pad(r, (0, 12 - len(r)), constant_values=-9))
    pairs_h.append(np.pad(h, (0, 12 - len(h)), constant_values=-9))
    vals.append(wer(list(r), list(h)))
out["pair_r"], out["pair_h"], out["pair_wer"] = np.asarray(pairs_r), np.asarray(pairs_h), np.asarray(vals)
# (ii) batches as main.py:43,201 feeds them: ignore [0,-1]; rows padded with -1; 0 = space label
calc = WERCalculator([0, -1])
B, W = 32, 10
lab = rng.integers(0, 5, (B, W))
hyp = rng.integers(0, 5, (B, W))
for b in range(B):
    lab[b, rng.integers(0, W + 1):] = -1
    hyp[b, rng.integers(0, W + 1):] = -1
lab[3] = -1                                   # empty target -> 0.0
hyp[5] = lab[5]
out["batch_r"], out["batch_h"] = lab, hyp
out["batch_wer"] = calc.cal_batch_wer(lab, hyp)
out["residual"] = np.asarray([len(calc.remove_residual(row)) for row in lab])
# (iii) top-k layout: 2 "gpus" x batch 4 x max_topk 3, use topk 2
r2 = rng.integers(1, 5, (8, 6))
h2 = rng.integers(1, 5, (2 * 4 * 3, 6))
out["topk_r"], out["topk_h"] = r2, h2
out["topk_wer"] = np.asarray(calc.cal_topk_wers(r2, h2, 4, 2, 2, 3))
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "wer_golden.npz"), **out)
print("wrote wer_golden.npz", {k: v.shape for k, v in out.items()})
