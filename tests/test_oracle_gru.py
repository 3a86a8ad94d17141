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
