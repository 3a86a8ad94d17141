"""int8 ("octbit") variant of the GRU path (BASELINE configs[2]) through the C ABI, against
oracle/gru_oracle.py:gru_forward_octbit (whose op arithmetic is pinned by the reference's known-answer tests).

The integer arithmetic must be exact.  What cannot be bit-identical are the fp32 pieces around it (MFMA layer 0,
the v_exp/v_rcp sigmoid and tanh, <= 3e-7): a 1e-7 change of an activation that sits on a rounding boundary of
the u8 quantiser moves one q by 1, i.e. one pre-activation by |w_q| * scale (<= 8.5e-4 with glorot weights).  So:
  * single-step tests feed the oracle the GPU's own fp32 inputs and demand near-exact results, bit-exact logits;
  * multi-frame tests bound the error statistically (quantiser flips are rare and do not grow with T)."""
import numpy as np
import pytest
import torch

from oracle import decode_oracle as D
from oracle import gru_oracle as G
from oracle import octbit_oracle as Q

pytestmark = pytest.mark.gpu


def _model(w, n_mel=40, layers=2, **kw):
    from keyword_spotting_amd import get_config
    from keyword_spotting_amd.rnn_ctc import DeployModel
    return DeployModel(get_config(n_mel=n_mel, num_layers=layers, precision="int8", **kw), w)


def _octbit_cell(lay, x, h):
    """Layer >= 1 cell of the quantised graph on given fp32 inputs (float32 numpy)."""
    f32 = np.float32
    gq, gs, gb = Q.octize_weight_int8_signed(lay["Wg"])
    cq, cs, cb = Q.octize_weight_int8_signed(lay["Wc"])
    g = G._sigmoid((Q.octbit_rows(np.concatenate([x, h], 1), gq, gs, gb) + lay["bg"]).astype(f32))
    r, u = g[:, :128], g[:, 128:]
    c = np.tanh((Q.octbit_rows(np.concatenate([x, (r * h).astype(f32)], 1), cq, cs, cb) + lay["bc"]).astype(f32))
    return (u * h + (f32(1) - u) * c).astype(f32)


@pytest.mark.parametrize("wscale,batch", [(1.0, 64), (8.0, 37)])
def test_single_step_is_exact_given_the_same_fp32_inputs(wscale, batch):
    """T = 1: layer 1 sees x = new layer-0 state and the projection sees h' = new layer-1 state, both returned by
    the call -- so the oracle can be fed the very same floats.  wscale = 8 makes most pair sums saturate."""
    w = G.random_weights(40, 128, 2, 6, seed=301)
    w["layers"][1]["Wg"] = (w["layers"][1]["Wg"] * wscale).astype(np.float32)
    w["layers"][1]["Wc"] = (w["layers"][1]["Wc"] * wscale).astype(np.float32)
    mel = G.synthetic_mel(batch, 1, 40, seed=302)
    st0 = (0.5 * np.random.default_rng(303).standard_normal((2, batch, 128))).astype(np.float32)
    r = _model(w).forward(torch.from_numpy(mel), torch.from_numpy(st0))
    got_s, got_l = r["state"].cpu().numpy(), r["logits"].cpu().numpy()
    want_h1 = _octbit_cell(w["layers"][1], got_s[0], st0[1])
    err = np.abs(got_s[1] - want_h1)
    assert np.mean(err < 2e-6) > 0.99 and err.max() < 5e-3 * wscale, (np.mean(err < 2e-6), err.max())
    # projection: one call per stream on its [1, H] block -> bit-exact
    fq, fs, fb = Q.octize_weight_int8_signed(w["Wfc"])
    want_l = (Q.octbit_rows(got_s[1], fq, fs, fb) + w["bfc"]).astype(np.float32)
    np.testing.assert_array_equal(got_l[:, 0, :], want_l)


@pytest.mark.parametrize("n_mel,layers,batch,frames", [(40, 2, 37, 40), (60, 2, 16, 25), (40, 1, 5, 20), (40, 3, 18, 12)])
def test_int8_matches_oracle(n_mel, layers, batch, frames):
    w = G.random_weights(n_mel, 128, layers, 6, seed=311)
    mel = G.synthetic_mel(batch, frames, n_mel, seed=312)
    st0 = (0.5 * np.random.default_rng(313).standard_normal((layers, batch, 128))).astype(np.float32)
    want_l, want_s = G.gru_forward_octbit(w, mel, st0)
    r = _model(w, n_mel, layers).forward(torch.from_numpy(mel), torch.from_numpy(st0))
    got_l, got_s = r["logits"].cpu().numpy(), r["state"].cpu().numpy()
    el, es = np.abs(got_l - want_l), np.abs(got_s - want_s)
    print("int8 vs oracle (L=%d): logits mean %.2e max %.2e | state mean %.2e max %.2e" % (layers, el.mean(), el.max(), es.mean(), es.max()))
    assert el.mean() < 2e-3 and el.max() < 0.1, (el.mean(), el.max())
    assert es.mean() < 2e-4 and es.max() < 2e-2, (es.mean(), es.max())
    np.testing.assert_allclose(r["softmax"].cpu().numpy(), G.softmax(got_l), atol=2e-6)


def test_int8_state_streams_across_chunks_and_projection_range_is_per_call():
    """The GRU state does not care how the frames are chunked (bitwise); the projection's activation range is per
    call as in the reference (one OctbitMatMul on the call's [T,H] block), so logits of a chunk equal the
    one-shot logits only through the oracle's per-call semantics."""
    w = G.random_weights(40, 128, 2, 6, seed=321)
    b, t = 19, 66
    mel = torch.from_numpy(G.synthetic_mel(b, t, 40, seed=322)).cuda()
    m = _model(w)
    one = m.forward(mel, m.zero_state(b))
    st = m.zero_state(b)
    pos, parts = 0, []
    for n in (21, 22, 23):
        r = m.forward(mel[:, pos:pos + n].contiguous(), st)
        st, pos = r["state"], pos + n
        parts.append(r["logits"].cpu().numpy())
    assert torch.equal(st, one["state"])
    want_first, _ = G.gru_forward_octbit(w, mel[:, :21].cpu().numpy())
    assert np.abs(parts[0] - want_first).mean() < 2e-3


def test_int8_sequence_length_reset_and_ragged_batch():
    w = G.random_weights(40, 128, 2, 6, seed=331)
    b, t = 21, 30                                   # 21 streams: a ragged last group
    mel = G.synthetic_mel(b, t, 40, seed=332)
    st0 = (0.5 * np.random.default_rng(333).standard_normal((2, b, 128))).astype(np.float32)
    seq = np.random.default_rng(334).integers(0, t + 1, b).astype(np.int32)
    seq[0], seq[1] = 0, t
    want_l, want_s = G.gru_forward_octbit(w, mel, st0, seq_len=seq)
    m = _model(w)
    r = m.forward(torch.from_numpy(mel), torch.from_numpy(st0), seq_len=torch.from_numpy(seq))
    got_l, got_s = r["logits"].cpu().numpy(), r["state"].cpu().numpy()
    assert np.abs(got_l - want_l).mean() < 2e-3 and np.abs(got_s - want_s).mean() < 2e-4
    # finished frames: zero output row -> logits == bias exactly; stream 0 never ran -> state untouched
    for k in range(b):
        np.testing.assert_array_equal(got_l[k, seq[k]:], np.broadcast_to(w["bfc"], (t - seq[k], 6)))
    np.testing.assert_array_equal(got_s[:, 0], st0[:, 0])
    # reset mask == zero state
    mask = np.zeros(b, np.uint8); mask[::2] = 1
    z = st0.copy(); z[:, mask.astype(bool)] = 0
    a = m.forward(torch.from_numpy(mel), torch.from_numpy(st0), reset_mask=torch.from_numpy(mask))
    c = m.forward(torch.from_numpy(mel), torch.from_numpy(z))
    assert torch.equal(a["logits"], c["logits"]) and torch.equal(a["state"], c["state"])


def test_int8_fused_tokens_follow_decode2_and_carry_prev_word():
    from keyword_spotting_amd.prediction import tokens_to_seq
    w = G.random_weights(40, 128, 2, 6, seed=341)
    w["Wfc"] = (w["Wfc"] * 3).astype(np.float32)       # peaky softmax -> words actually fire
    b, t = 24, 70                                      # 70 frames: three 32-frame projection blocks with halos
    mel = torch.from_numpy(G.synthetic_mel(b, t, 40, seed=342)).cuda()
    m = _model(w)
    pw = m.fresh_prev_word(b)
    r = m.forward(mel, m.zero_state(b), prev_word=pw)
    sm, toks = r["softmax"].cpu().numpy(), r["tokens"].cpu().numpy()
    fired = 0
    for k in range(b):
        words = D.frame_words(sm[k], 1, 5, 0.4)
        margin = np.abs(np.sort(sm[k][:, 1:5], axis=1)[:, -1] - 0.4).min()
        if margin < 1e-5:
            continue
        np.testing.assert_array_equal(tokens_to_seq(toks[k]), D.ctc_decode2(sm[k], 6))
        assert int(pw[k]) == int(words[-1])
        fired += int((toks[k] > 0).sum())
    assert fired > 0
    # carry: a second call continues the neighbour rule from the last word of the first
    st = m.zero_state(b)
    pw2 = m.fresh_prev_word(b)
    r1 = m.forward(mel[:, :33].contiguous(), st, prev_word=pw2)
    r2 = m.forward(mel[:, 33:].contiguous(), r1["state"], prev_word=pw2)
    for k in range(b):
        w1 = D.frame_words(r1["softmax"][k].cpu().numpy(), 1, 5, 0.4)
        w2 = D.frame_words(r2["softmax"][k].cpu().numpy(), 1, 5, 0.4)
        first = int(r2["tokens"][k, 0])
        assert first == (int(w2[0]) + 1 if (w2[0] >= 0 and w2[0] != w1[-1]) else 0)


def test_int8_versus_fp32_reference_semantics_report():
    """Config 3 report: error of the quantised graph against the fp32 graph, GPU and oracle side by side."""
    w = G.init_weights()
    b, t = 32, 100
    mel = G.synthetic_mel(b, t, 40, seed=351)
    ref_l, _ = G.gru_forward(w, mel, dtype=np.float64)
    want_l, _ = G.gru_forward_octbit(w, mel)
    got = _model(w).forward(torch.from_numpy(mel), torch.zeros(2, b, 128))["logits"].cpu().numpy()
    e_gpu, e_or = np.abs(got - ref_l), np.abs(want_l - ref_l)
    print("int8 vs fp32 semantics: GPU mean|dlogit| %.3f max %.3f ; oracle (reference int8 arithmetic) mean %.3f max %.3f"
          % (e_gpu.mean(), e_gpu.max(), e_or.mean(), e_or.max()))
    assert abs(e_gpu.mean() - e_or.mean()) < 0.02 * max(1.0, e_or.mean())


def test_int8_unsupported_shapes_are_refused():
    from keyword_spotting_amd import _lib
    w = G.random_weights(40, 256, 2, 6, seed=361)
    with pytest.raises(_lib.UnsupportedError):
        from keyword_spotting_amd import get_config
        from keyword_spotting_amd.rnn_ctc import DeployModel
        DeployModel(get_config(hidden_size=256, precision="int8"), w)
