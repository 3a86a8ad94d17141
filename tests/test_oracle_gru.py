"""GRU oracle: PARITY UNPINNED by the reference (TF 1.x absent, no reference test at this stage), so
the restatement is cross-checked three independent ways plus its structural invariants."""
import numpy as np
import pytest
import torch

from oracle import gru_oracle as G
from oracle import torch_eager as TE


def test_three_formulations_agree():
    w = G.random_weights(40, 128, 2, 6, seed=3)
    mel = G.synthetic_mel(3, 40, 40, seed=4)
    l64, s64 = G.gru_forward(w, mel, dtype=np.float64)
    l32, s32 = G.gru_forward(w, mel, dtype=np.float32)
    lsp, ssp = G.gru_forward_split(w, mel, dtype=np.float64)
    np.testing.assert_allclose(lsp, l64, atol=1e-12)
    np.testing.assert_allclose(l32, l64, atol=2e-5)
    np.testing.assert_allclose(s32, s64, atol=2e-6)
    tw = TE.to_torch(w, torch.float64)
    lt, smt, st = TE.gru_forward(tw, torch.from_numpy(mel).double(), torch.zeros(2, 3, 128, dtype=torch.float64))
    np.testing.assert_allclose(lt.numpy(), l64, atol=1e-11)
    np.testing.assert_allclose(st.numpy(), s64, atol=1e-12)
    np.testing.assert_allclose(smt.numpy(), G.softmax(l64), atol=1e-12)


def test_gate_order_and_reset_before_matmul():
    """TF GRUCell != cuDNN/torch.nn.GRU: r multiplies h BEFORE the candidate matmul, gates are [r,u],
    h' = u*h + (1-u)*c.  Hand-computed single unit."""
    w = dict(layers=[dict(Wg=np.array([[0.5, -1.0], [2.0, 0.25]], np.float64), bg=np.array([0.1, -0.2]),
                          Wc=np.array([[1.5], [-0.75]], np.float64), bc=np.array([0.05]))],
             Wfc=np.array([[1.0, 0, 0]], np.float64), bfc=np.zeros(3))
    x, h = 0.3, -0.4
    r = 1 / (1 + np.exp(-(0.5 * x + 2.0 * h + 0.1)))
    u = 1 / (1 + np.exp(-(-1.0 * x + 0.25 * h - 0.2)))
    c = np.tanh(1.5 * x - 0.75 * (r * h) + 0.05)
    want = u * h + (1 - u) * c
    lg, st = G.gru_forward(w, np.array([[[x]]]), np.array([[[h]]]), dtype=np.float64)
    assert abs(st[0, 0, 0] - want) < 1e-15 and abs(lg[0, 0, 0] - want) < 1e-15


@pytest.mark.parametrize("splits", [[300], [21, 23, 22, 23, 22, 23, 22, 23, 22, 99], [1] * 12 + [288]])
def test_chunked_equals_one_shot(splits):
    """detector.py test2 idea (:254-289): any split with carried state == single call."""
    w = G.init_weights()
    mel = G.synthetic_mel(2, 300)
    whole, s_whole = G.gru_forward(w, mel)
    chunked, s_chunk = G.stream_chunks(w, mel, splits)
    np.testing.assert_array_equal(s_chunk, s_whole)          # recurrence: bitwise
    # the dense layer runs as one [B*T,H] BLAS call whose blocking depends on T: last-bit only
    np.testing.assert_allclose(chunked, whole, atol=2e-6, rtol=0)


def test_sequence_length_copy_through():
    w = G.random_weights(40, 128, 2, 6, seed=5)
    mel = G.synthetic_mel(3, 20)
    lens = np.array([20, 7, 0])
    lg, st = G.gru_forward(w, mel, seq_len=lens)
    for b, n in enumerate(lens):
        lg_b, st_b = G.gru_forward(w, mel[b:b + 1, :n])
        np.testing.assert_allclose(lg[b, :n], lg_b[0], atol=1e-5, rtol=0)   # BLAS blocking differs with B
        np.testing.assert_allclose(st[:, b], st_b[:, 0], atol=1e-6, rtol=0)
        np.testing.assert_array_equal(lg[b, n:], np.broadcast_to(w["bfc"], (20 - n, 6)))


def test_relu_clip():
    w = G.random_weights(40, 128, 2, 6, seed=6)
    w["Wfc"] *= 20
    mel = G.synthetic_mel(1, 30)
    raw, _ = G.gru_forward(w, mel)
    relu, _ = G.gru_forward(w, mel, use_relu=True, value_clip=1.0)
    np.testing.assert_array_equal(relu, np.clip(raw, 0, 20))
    assert (raw < 0).any() and (raw > 20).any()


@pytest.mark.parametrize("shape", [(40, 128, 2, 6), (60, 128, 2, 6), (60, 256, 4, 6), (13, 64, 1, 4)])
def test_c_oracle_matches_numpy(oracle_c, shape):
    i, h, l, c = shape
    w = G.random_weights(i, h, l, c, seed=7)
    mel = G.synthetic_mel(5, 33, i, seed=8)
    rng = np.random.default_rng(9)
    st0 = (0.5 * rng.standard_normal((l, 5, h))).astype(np.float32)
    lens = np.array([33, 0, 12, 33, 1], np.int32)
    want_l, want_s = G.gru_forward(w, mel, st0, seq_len=lens, dtype=np.float64)
    got_l, got_sm, got_s = oracle_c.gru_forward((i, h, l, c, 0, -1.0), G.weights_to_blob(w), mel, st0, lens, threads=2)
    np.testing.assert_allclose(got_l, want_l, atol=3e-5)
    np.testing.assert_allclose(got_s, want_s, atol=5e-6)
    np.testing.assert_allclose(got_sm, G.softmax(want_l), atol=5e-6)


def test_blob_size_config_a():
    assert G.weights_to_blob(G.init_weights()).nbytes == 657432   # 164,358 params


def test_what_single_piece_fp16_operands_would_cost():
    """an `f16x1` variant of gru_layer_f16x3 -- ONE fp16 piece per operand instead of two: 49 instead of 147
    MFMAs and ~110 instead of ~170 VALU instructions per frame and wave, hi-only weights (321 KiB) that fit one launch -- with the
    stop rule "keep only if >= 0.9 of the streams keep their fp32 token sequence over 300 frames".  Priced on the rounding model
    before any kernel was written (the same model, with bf16 roundings, predicts what the bf16 HIP stack measures: 0.30-0.33 of
    the streams in bench.py's `accuracy` entry, 0.32 here at 256 streams):

        operands                      256 streams x 300 frames, random-init weights      streams with the fp32 token sequence
        bf16 weights, bf16 inputs     (configs[2])                                        0.32
        fp16 weights, fp16 inputs     (f16x1)                                             0.77
        fp16 weights, full inputs     (2 MFMAs per product)                               0.77
        full weights, fp16 inputs     (2 MFMAs per product, no activation split)          0.90

    The weights' static 2^-12 perturbation alone loses a quarter of the streams on this (undecided, random-init) model: f16x1
    fails its rule at 0.77, and the one 2-MFMA form that reaches 0.90 would be ~1.4x f16x3 (98 MFMAs, ~140 VALU), below the
    1.5x the rule also asks.  ON THE HARDWARE the "fp16 weights, full inputs" row was then measured with the f16x3 kernels and
    the lo piece of every weight zeroed (tools/exp_f16_single_piece.py on a -DKWS_EXP_F16_WLO_ZERO variant build,
    profiles/r6_f16_single_piece.txt): 0.55 of 256 streams (max |dlogit| 5.5e-3; the kernels' weights carry the folded exponent
    scales, whose rounding the model does not have), bf16 0.29 on the same inputs.  Not built.  This test
    keeps the ordering those numbers rest on, at a size that runs in seconds."""
    w = G.init_weights(seed=0)
    b, t = 48, 300
    mel = (np.abs(np.random.default_rng(5).standard_normal((b, t, 40))) * 2).astype(np.float32)
    ident = lambda a: np.asarray(a, np.float32)

    def tokens(logits):
        sm = G.softmax(logits)
        p = sm[:, :, 1:5]
        word = np.where(p.max(-1) > 0.4, p.argmax(-1), -1)
        prev = np.concatenate([np.full((b, 1), -1), word[:, :-1]], 1)
        return np.where((word >= 0) & (word != prev), word + 1, 0)

    ref = tokens(G.gru_forward_rounded(w, mel, ident, ident)[0])
    same = {}
    for name, rw, rx in (("bf16", G.bf16_round, G.bf16_round), ("f16x1", G.f16_round, G.f16_round),
                         ("w16", G.f16_round, ident), ("x16", ident, G.f16_round)):
        got = tokens(G.gru_forward_rounded(w, mel, rw, rx)[0])
        same[name] = ((got == ref).all(1).mean(), (got == ref).mean())
    assert int((ref > 0).sum()) > 10 * b                          # the model emits ~30 words per stream: the comparison is not vacuous
    assert same["bf16"][0] < same["f16x1"][0] < 0.9               # eight times finer than bf16, still short of the rule
    assert same["x16"][0] >= same["w16"][0]                       # the weights' rounding is what costs the streams
    assert same["f16x1"][1] > 0.995 and same["bf16"][1] > 0.98   # frame by frame nearly everything agrees: the rule is per 300-frame stream
    np.testing.assert_allclose(G.gru_forward_rounded(w, mel[:4, :20], G.bf16_round, G.bf16_round)[0], G.gru_forward_bf16(w, mel[:4, :20])[0], atol=1e-12)
