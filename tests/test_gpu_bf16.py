"""bf16 variant (BASELINE configs[2]) of the GRU path: against an oracle that rounds where the kernel
rounds, and against the fp32 reference semantics (error and token-agreement report)."""
import numpy as np
import pytest
import torch

from oracle import decode_oracle as D
from oracle import gru_oracle as G

pytestmark = pytest.mark.gpu


def _model(w, n_mel=40, layers=2, **kw):
    from keyword_spotting_amd import get_config
    from keyword_spotting_amd.rnn_ctc import DeployModel
    return DeployModel(get_config(n_mel=n_mel, num_layers=layers, precision="bf16", **kw), w)


@pytest.mark.parametrize("n_mel,layers,batch,frames", [(40, 2, 37, 50), (60, 2, 16, 33), (40, 1, 5, 20), (32, 2, 3, 300)])
def test_bf16_matches_rounding_oracle(n_mel, layers, batch, frames):
    w = G.random_weights(n_mel, 128, layers, 6, seed=101)
    mel = G.synthetic_mel(batch, frames, n_mel, seed=102)
    st0 = (0.5 * np.random.default_rng(103).standard_normal((layers, batch, 128))).astype(np.float32)
    want_l, want_s = G.gru_forward_bf16(w, mel, st0)
    r = _model(w, n_mel, layers).forward(torch.from_numpy(mel), torch.from_numpy(st0))
    got_l, got_s = r["logits"].cpu().numpy(), r["state"].cpu().numpy()
    # a rounding of h / r*h that lands on the other side of a bf16 tie costs 2^-8 relative on that value
    # (stationary, does not grow with T: measured max 0.02 / mean 0.003 at T=300)
    assert np.abs(got_l - want_l).max() < 6e-2 and np.abs(got_l - want_l).mean() < 6e-3
    assert np.abs(got_s - want_s).max() < 2e-2 and np.abs(got_s - want_s).mean() < 1e-3
    np.testing.assert_allclose(r["softmax"].cpu().numpy().sum(-1), 1.0, atol=1e-6)


def test_bf16_versus_fp32_reference_semantics():
    """Config 3 report: logits error and token-sequence agreement of the bf16 path vs the fp32 semantics."""
    w = G.init_weights()
    b, t = 64, 300
    mel = G.synthetic_mel(b, t, 40, seed=111)
    want_l, _ = G.gru_forward(w, mel, dtype=np.float64)
    m = _model(w)
    pw = m.fresh_prev_word(b)
    r = m.forward(torch.from_numpy(mel), m.zero_state(b), prev_word=pw)
    got = r["logits"].cpu().numpy()
    err = np.abs(got - want_l)
    assert err.mean() < 2e-2 and err.max() < 0.5, (err.mean(), err.max())
    from keyword_spotting_amd.prediction import tokens_to_seq
    sm = G.softmax(want_l)
    toks = r["tokens"].cpu().numpy()
    agree = sum(np.array_equal(tokens_to_seq(toks[k]), D.ctc_decode2(sm[k], 6)) for k in range(b))
    frame_agree = np.mean([(D.frame_words(G.softmax(got[k]), 1, 5, 0.4) == D.frame_words(sm[k], 1, 5, 0.4)).mean() for k in range(b)])
    print("bf16 vs fp32: mean|dlogit|=%.2e max=%.2e  identical token sequences %d/%d  frame-word agreement %.4f"
          % (err.mean(), err.max(), agree, b, frame_agree))
    assert frame_agree > 0.99


def test_bf16_chunked_equals_one_shot_and_masks():
    w = G.random_weights(40, 128, 2, 6, seed=121)
    b, t = 19, 90
    mel = torch.from_numpy(G.synthetic_mel(b, t, 40, seed=122)).cuda()
    m = _model(w)
    whole = m.forward(mel, m.zero_state(b))
    state, outs, pos = m.zero_state(b), [], 0
    for n in (21, 22, 23, 1, 23):
        lg, state = m.step(mel[:, pos:pos + n].contiguous(), state)
        outs.append(lg)
        pos += n
    assert torch.equal(torch.cat(outs, 1), whole["logits"]) and torch.equal(state, whole["state"])
    rng = np.random.default_rng(123)
    lens = rng.integers(0, t + 1, b).astype(np.int32)
    reset = (rng.random(b) < 0.4).astype(np.uint8)
    st0 = (0.5 * rng.standard_normal((2, b, 128))).astype(np.float32)
    want_l, want_s = G.gru_forward_bf16(w, mel.cpu().numpy(), st0 * (1 - reset)[None, :, None], seq_len=lens)
    r = m.forward(mel, torch.from_numpy(st0), seq_len=torch.from_numpy(lens), reset_mask=torch.from_numpy(reset))
    assert np.abs(r["logits"].cpu().numpy() - want_l).max() < 5e-2
    assert np.abs(r["state"].cpu().numpy() - want_s).max() < 2e-2
    got = r["logits"].cpu().numpy()
    for k in range(b):
        np.testing.assert_array_equal(got[k, lens[k]:], np.broadcast_to(w["bfc"], (t - lens[k], 6)))


def test_bf16_unsupported_shapes_fail_loudly():
    from keyword_spotting_amd import _lib
    with pytest.raises(_lib.UnsupportedError):
        _model(G.init_weights(60, 256, 2, 6), n_mel=60, hidden_size=256)
    with pytest.raises(_lib.UnsupportedError):
        _model(G.init_weights(40, 128, 4, 6), layers=4)
