"""The validation metrics of main.py:186-217 fed by the GPU path: softmax from the HIP GRU stack, ctc_decode and
ctc_predict on the device, then the host-side `evaluate` (utils/prediction.py:203-210) and `WERCalculator`
(utils/wer.py:80-106) -- against the same loop run entirely on the oracle (fp64 GRU, reference-pinned decoders).
evaluate / WER are host numpy by design (they run once per validation batch on a handful of label rows)."""
import numpy as np
import pytest
import torch

from oracle import decode_oracle as D
from oracle import gru_oracle as G

pytestmark = pytest.mark.gpu


def test_validation_loop_metrics_match_the_oracle_loop():
    from keyword_spotting_amd import get_config
    from keyword_spotting_amd.prediction import ctc_decode, ctc_predict, evaluate
    from keyword_spotting_amd.rnn_ctc import DeployModel
    from keyword_spotting_amd.wer import WERCalculator
    w = G.init_weights(seed=4)
    w["Wfc"] = (w["Wfc"] * 3.0).astype(np.float32)                   # peaky softmax: words actually fire
    b, t = 48, 120
    mel = G.synthetic_mel(b, t, 40, seed=601)
    rng = np.random.default_rng(602)
    seq_len = rng.integers(40, t + 1, b).astype(np.int32)            # padded validation batch (models/rnn_ctc.py:50-57)
    label = None                                                      # picked below so that it fires on at least one row
    correctness = rng.integers(0, 2, b)                               # main.py: 1 = the utterance holds the keyword
    labels = [np.concatenate((rng.integers(1, 4, int(rng.integers(1, 6))), [-1, -1])) for _ in range(b)]

    m = DeployModel(get_config(), w)
    r = m.forward(torch.from_numpy(mel), m.zero_state(b), seq_len=torch.from_numpy(seq_len))
    softmax = r["softmax"]
    want_l, _ = G.gru_forward(w, mel, seq_len=seq_len, dtype=np.float64)
    want_sm = G.softmax(want_l)

    def clear_row(k):
        n = int(seq_len[k])
        p = np.sort(want_sm[k, :n, 1:5], axis=1)
        return bool((np.abs(p[:, -1] - 0.5) > 1e-4).all() and (np.abs(p[:, -1] - 0.2) > 1e-4).all() and
                    (np.abs(want_sm[k, :n, 3] - 0.2) > 1e-4).all() and (np.abs(p[:, -1] - 0.6) > 1e-4).all() and
                    ((p[:, -1] - p[:, -2] > 1e-4) | (p[:, -1] < 0.15)).all())   # ties only matter above a threshold

    rows = [k for k in range(b) if clear_row(k)]                      # no frame within 1e-4 of a threshold or tie
    refs = {k: D.ctc_decode(want_sm[k, :int(seq_len[k])]) for k in rows}
    for k in rows:                                                    # a label some clip really spells (random weights rarely say "1233")
        if label is None and len(refs[k][1::2]) >= 2:
            label = "%d%d" % (refs[k][1], refs[k][3])
    assert label is not None
    calc = WERCalculator([0, -1])
    got_res, want_res, got_wer, want_wer, checked = [], [], [], [], 0
    for k in rows:
        dec = ctc_decode(softmax[k, :int(seq_len[k])])               # device decode, reference format [0,w,0,...]
        np.testing.assert_array_equal(np.asarray(dec), refs[k])
        got_res.append(ctc_predict(dec, label)); want_res.append(D.ctc_predict(refs[k], label))
        got_wer.append(calc.cal_batch_wer([labels[k]], [np.asarray(dec)])[0])
        want_wer.append(calc.cal_batch_wer([labels[k]], [refs[k]])[0])
        checked += 1
    assert checked >= b // 2
    target = correctness[:checked].tolist()
    assert evaluate(got_res, target) == D.evaluate(want_res, target)
    assert sum(got_res) > 0                                           # the keyword does fire on some rows
    np.testing.assert_array_equal(np.asarray(got_wer), np.asarray(want_wer))
