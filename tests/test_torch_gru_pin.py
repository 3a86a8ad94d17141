"""Third-party pin of the GRU STACK over whole sequences: torch.nn.GRU (PyTorch, written by neither the reference's author nor
this repo's) on the sub-family of weights where it and TensorFlow's GRUCell are the same function.

The two cells differ in ONE place: TF-1.x GRUCell (the reference's cell, models/rnn_ctc.py:179-185) applies the reset gate BEFORE
the candidate's recurrent matmul, c = tanh(x Wx + (r (.) h) Wh + b); PyTorch / cuDNN apply it AFTER, n = tanh(x Wx + b + r (.) (h Wh)).
For a DIAGONAL Wh the two coincide (r (.) (h D) = (r (.) h) D); everything else -- gate order, gate biases, the update
h' = u h + (1 - u) c, layer stacking, state carry over the sequence, the dense layer -- is exercised with full random weights.
The reset-before-matmul rule itself is the ONE element no third-party number available here distinguishes: TensorFlow's published
constants (tests/test_tf_published_kat.py) use two symmetric units, for which both conventions give 0.175991.  It rests on the TF-1.x
source as restated in SURVEY.md R4 / oracle/gru_oracle.py (`candidate = linear([inputs, r * state])`) and on the hand-computed
asymmetric case tests/test_oracle_gru.py::test_gate_order_and_reset_before_matmul; test_the_two_cells_really_differ... below shows
that the kernels do NOT follow the PyTorch/cuDNN convention.

Not the reference run here, so the GRU stage stays "parity partial" (DESIGN.md section 5)."""
import numpy as np
import pytest
import torch

from oracle import gru_oracle as G


def diag_candidate_weights(n_mel, hidden, layers, classes, seed):
    """Random TF-layout weights whose candidate-recurrent block Wc[I:, :] is diagonal."""
    w = G.random_weights(n_mel, hidden, layers, classes, seed)
    rng = np.random.default_rng(seed + 1)
    for l, lay in enumerate(w["layers"]):
        i_l = n_mel if l == 0 else hidden
        lay["Wc"] = lay["Wc"].copy()
        lay["Wc"][i_l:, :] = np.diag(rng.uniform(-0.9, 0.9, hidden)).astype(np.float32)
    return w


def torch_stack(w, n_mel, hidden):
    """torch.nn.GRU (+ Linear) carrying the same function as the TF-layout weights `w` (float64)."""
    layers = len(w["layers"])
    gru = torch.nn.GRU(n_mel, hidden, num_layers=layers, batch_first=True).double()
    with torch.no_grad():
        for l, lay in enumerate(w["layers"]):
            i_l = n_mel if l == 0 else hidden
            wg, wc = lay["Wg"].astype(np.float64), lay["Wc"].astype(np.float64)
            # torch rows: (r, z, n); TF columns of Wg: (r | u), u plays z
            w_ih = np.concatenate([wg[:i_l, :hidden].T, wg[:i_l, hidden:].T, wc[:i_l].T], 0)
            w_hh = np.concatenate([wg[i_l:, :hidden].T, wg[i_l:, hidden:].T, wc[i_l:].T], 0)
            b_ih = np.concatenate([lay["bg"][:hidden], lay["bg"][hidden:], lay["bc"]]).astype(np.float64)
            getattr(gru, "weight_ih_l%d" % l).copy_(torch.from_numpy(w_ih))
            getattr(gru, "weight_hh_l%d" % l).copy_(torch.from_numpy(w_hh))
            getattr(gru, "bias_ih_l%d" % l).copy_(torch.from_numpy(b_ih))
            getattr(gru, "bias_hh_l%d" % l).zero_()
    fc_w, fc_b = torch.from_numpy(w["Wfc"].astype(np.float64)), torch.from_numpy(w["bfc"].astype(np.float64))

    @torch.no_grad()
    def run(mel, state, lens=None):
        x, h0 = torch.from_numpy(np.asarray(mel, np.float64)), torch.from_numpy(np.asarray(state, np.float64))
        if lens is None:
            out, hn = gru(x, h0)
        else:
            # packed sequences: rows past a stream's length come back as zeros and h_n is the state at its last valid frame --
            # what dynamic_rnn(sequence_length=...) does (models/rnn_ctc.py:238-243)
            pk = torch.nn.utils.rnn.pack_padded_sequence(x, torch.as_tensor(lens, dtype=torch.int64), batch_first=True, enforce_sorted=False)
            out, hn = gru(pk, h0)
            out, _ = torch.nn.utils.rnn.pad_packed_sequence(out, batch_first=True, total_length=x.shape[1])
        return (out @ fc_w + fc_b).numpy(), hn.numpy()
    return run


@pytest.mark.parametrize("n_mel,hidden,layers", [(40, 128, 2), (60, 64, 3), (7, 16, 1)])
def test_oracle_equals_torch_gru_on_the_common_family(n_mel, hidden, layers):
    w = diag_candidate_weights(n_mel, hidden, layers, 6, seed=71)
    mel = G.synthetic_mel(3, 60, n_mel, seed=72)
    st0 = (0.5 * np.random.default_rng(73).standard_normal((layers, 3, hidden))).astype(np.float32)
    want_l, want_s = torch_stack(w, n_mel, hidden)(mel, st0)
    for fwd in (G.gru_forward, G.gru_forward_split):
        got_l, got_s = fwd(w, mel, st0, dtype=np.float64)
        np.testing.assert_allclose(got_l, want_l, atol=1e-11)
        np.testing.assert_allclose(got_s, want_s, atol=1e-12)


def test_sequence_lengths_match_torch_packed_sequences():
    """dynamic_rnn(sequence_length): zero output rows (logits = bias) and state copy-through past a stream's length."""
    w = diag_candidate_weights(40, 128, 2, 6, seed=91)
    b, t = 9, 30
    mel = G.synthetic_mel(b, t, 40, seed=92)
    st0 = (0.4 * np.random.default_rng(93).standard_normal((2, b, 128))).astype(np.float32)
    lens = np.array([30, 1, 17, 29, 2, 30, 8, 15, 23])
    want_l, want_s = torch_stack(w, 40, 128)(mel, st0, lens)
    got_l, got_s = G.gru_forward(w, mel, st0, seq_len=lens, dtype=np.float64)
    np.testing.assert_allclose(got_l, want_l, atol=1e-11)
    np.testing.assert_allclose(got_s, want_s, atol=1e-12)


def test_the_two_cells_really_differ_off_the_family():
    """With a full candidate-recurrent matrix torch.nn.GRU is NOT the reference's cell (reset after vs before the matmul): the
    oracle follows TF's convention, and the pin above is not vacuous."""
    w = G.random_weights(40, 128, 1, 6, seed=74)
    mel = G.synthetic_mel(2, 20, 40, seed=75)
    st0 = np.zeros((1, 2, 128), np.float32)
    torch_l, _ = torch_stack(w, 40, 128)(mel, st0)
    ours_l, _ = G.gru_forward(w, mel, st0, dtype=np.float64)
    assert np.abs(torch_l - ours_l).max() > 1e-3


def test_c_oracle_equals_torch_gru_on_the_common_family(oracle_c):
    w = diag_candidate_weights(40, 128, 2, 6, seed=76)
    mel = G.synthetic_mel(4, 80, 40, seed=77)
    st0 = (0.3 * np.random.default_rng(78).standard_normal((2, 4, 128))).astype(np.float32)
    want_l, want_s = torch_stack(w, 40, 128)(mel, st0)
    c_l, _, c_s = oracle_c.gru_forward((40, 128, 2, 6, 0, -1.0), G.weights_to_blob(w), mel, st0)
    assert np.abs(c_l - want_l).max() < 2e-5 and np.abs(c_s - want_s).max() < 5e-6


# ---------------------------------------------------------------------------------------------- GPU, through the C ABI
@pytest.mark.gpu
@pytest.mark.parametrize("n_mel,hidden,layers,kernel,precision", [
    (40, 128, 2, "resident", "fp32"), (40, 128, 2, "generic", "fp32"), (40, 128, 2, "auto", "f16x3"),
    (60, 128, 3, "auto", "f16x3"), (60, 256, 2, "generic", "fp32"), (60, 64, 2, "generic", "fp32")])
def test_kws_step_equals_torch_gru_over_a_300_frame_sequence(n_mel, hidden, layers, kernel, precision):
    from keyword_spotting_amd import get_config
    from keyword_spotting_amd.rnn_ctc import DeployModel
    w = diag_candidate_weights(n_mel, hidden, layers, 6, seed=81)
    b, t = 21, 300
    mel = G.synthetic_mel(b, t, n_mel, seed=82)
    st0 = (0.5 * np.random.default_rng(83).standard_normal((layers, b, hidden))).astype(np.float32)
    want_l, want_s = torch_stack(w, n_mel, hidden)(mel, st0)
    m = DeployModel(get_config(n_mel=n_mel, hidden_size=hidden, num_layers=layers, precision=precision), w, kernel=kernel)
    r = m.forward(torch.from_numpy(mel), torch.from_numpy(st0))
    assert np.abs(r["logits"].cpu().numpy() - want_l).max() < 1e-4          # north_star's tolerance
    assert np.abs(r["state"].cpu().numpy() - want_s).max() < 1e-4
    sm = torch.softmax(torch.from_numpy(want_l), -1).numpy()
    assert np.abs(r["softmax"].cpu().numpy() - sm).max() < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("kernel,precision", [("resident", "fp32"), ("generic", "fp32"), ("auto", "f16x3")])
def test_kws_step_sequence_lengths_equal_torch_packed_sequences(kernel, precision):
    from keyword_spotting_amd import get_config
    from keyword_spotting_amd.rnn_ctc import DeployModel
    w = diag_candidate_weights(40, 128, 2, 6, seed=94)
    b, t = 37, 41
    mel = G.synthetic_mel(b, t, 40, seed=95)
    rng = np.random.default_rng(96)
    st0 = (0.4 * rng.standard_normal((2, b, 128))).astype(np.float32)
    lens = rng.integers(1, t + 1, b)
    lens[:3] = [t, 1, 2]
    want_l, want_s = torch_stack(w, 40, 128)(mel, st0, lens)
    m = DeployModel(get_config(precision=precision), w, kernel=kernel)
    r = m.forward(torch.from_numpy(mel), torch.from_numpy(st0), seq_len=torch.from_numpy(lens.astype(np.int32)))
    assert np.abs(r["logits"].cpu().numpy() - want_l).max() < 1e-4
    assert np.abs(r["state"].cpu().numpy() - want_s).max() < 1e-4
