"""WER metric mirror (keyword_spotting_amd/wer.py) against goldens produced by the reference's own
utils/wer.py (tests/golden/make_wer_golden.py)."""
import os

import numpy as np

from keyword_spotting_amd.wer import WERCalculator, edit_distance, wer

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "wer_golden.npz"))


def test_pairwise_wer_matches_reference():
    for r, h, v in zip(G["pair_r"], G["pair_h"], G["pair_wer"]):
        assert wer(r[r != -9], h[h != -9]) == v


def test_batch_wer_and_residual_removal():
    calc = WERCalculator([0, -1])
    np.testing.assert_array_equal(calc.cal_batch_wer(G["batch_r"], G["batch_h"]), G["batch_wer"])
    assert [len(calc.remove_residual(r)) for r in G["batch_r"]] == list(G["residual"])


def test_topk_layout():
    calc = WERCalculator([0, -1])
    got = calc.cal_topk_wers(G["topk_r"], G["topk_h"], 4, 2, 2, 3)
    np.testing.assert_array_equal(np.asarray(got), G["topk_wer"])


def test_distance_properties_beyond_reference_limit():
    rng = np.random.default_rng(0)
    a = rng.integers(0, 4, 600)
    assert edit_distance(a, a) == 0
    assert edit_distance(a, a[:-300]) == 300          # the reference's uint8 table would wrap here
    assert edit_distance([], a) == 600 and wer([], [1, 2]) == 2.0
    b = a.copy(); b[::50] = 9
    assert edit_distance(a, b) == len(a[::50])
