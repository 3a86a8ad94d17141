"""Third-party pin of the GRU cell: the constants TensorFlow 1.x publishes in its own unit tests
(tensorflow/python/kernel_tests/rnn_cell_test.py, TF 1.1 - 1.15, unchanged across those releases).

  testGRUCell        GRUCell(2), every variable initialised to 0.5 by the enclosing scope, the cell's own
                     bias initialisers (gates 1.0, candidate 0.0) taking precedence:
                       x=[[1,1]],   h=[[.1,.1]]  ->  [[0.175991, 0.175991]]
                       x=[[1,1,1]], h=[[.1,.1]]  ->  [[0.156736, 0.156736]]     (input_size != num_units)
  testMultiRNNCell   MultiRNNCell([GRUCell(2), GRUCell(2)]), x=[[1,1]], state=[[.1,.1,.1,.1]]
                       -> new state [[0.175991, 0.175991, 0.13248, 0.13248]]    (layer l feeds layer l+1)

These are what `tensorflow.contrib.rnn.GRUCell` / `MultiRNNCell` -- the cells models/rnn_ctc.py:179-185,228-236
instantiates -- return; they are not "the reference run here" (TF cannot be installed), so DESIGN.md keeps the GRU
stage at "parity partial", but they are evidence that does not come from this repo's author.

CPU part: the three oracle formulations.  GPU part: the same cells embedded in the shipped shapes (H = 64 / 128;
the live units' rows and columns 0.5, everything else 0 -> the other units stay exactly 0) through kws_step.
"""
import numpy as np
import pytest
import torch

from oracle import gru_oracle as G
from oracle import torch_eager as TE

KAT_2 = 0.175991      # x = [1,1]
KAT_3 = 0.156736      # x = [1,1,1]
KAT_L1 = 0.13248      # second GRUCell of the MultiRNNCell, fed 0.175991
TF_TOL = 1e-6         # assertAllClose on 6 printed digits


def tf_cell(n_in, units=2):
    return dict(Wg=np.full((n_in + units, 2 * units), 0.5, np.float32), bg=np.ones(2 * units, np.float32),
                Wc=np.full((n_in + units, units), 0.5, np.float32), bc=np.zeros(units, np.float32))


def embedded(n_mel, hidden, n_in, num_layers):
    """The 2-unit TF test cell(s) inside an [n_mel, hidden] stack: units 0,1 and inputs 0..n_in-1 are live."""
    layers = []
    for l in range(num_layers):
        i_l = n_mel if l == 0 else hidden
        live_in = n_in if l == 0 else 2
        wg = np.zeros((i_l + hidden, 2 * hidden), np.float32)
        wc = np.zeros((i_l + hidden, hidden), np.float32)
        rows = list(range(live_in)) + [i_l, i_l + 1]
        for r in rows:
            wg[r, [0, 1, hidden, hidden + 1]] = 0.5
            wc[r, [0, 1]] = 0.5
        layers.append(dict(Wg=wg, bg=np.ones(2 * hidden, np.float32), Wc=wc, bc=np.zeros(hidden, np.float32)))
    wfc = np.zeros((hidden, 6), np.float32)
    wfc[0, 0] = wfc[1, 1] = 1.0                      # logits[.., 0:2] = the live units of the top layer
    return dict(layers=layers, Wfc=wfc, bfc=np.zeros(6, np.float32))


def embedded_inputs(n_mel, hidden, n_in, num_layers, batch=1):
    mel = np.zeros((batch, 1, n_mel), np.float32)
    mel[:, 0, :n_in] = 1.0
    st = np.zeros((num_layers, batch, hidden), np.float32)
    st[:, :, :2] = 0.1
    return mel, st


# ---------------------------------------------------------------------------------------------- CPU
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("cell", [G.gru_cell, G.gru_cell_split])
def test_oracle_cell_reproduces_tf_testGRUCell(cell, dtype):
    h = np.array([[0.1, 0.1]], dtype)
    np.testing.assert_allclose(cell(np.ones((1, 2), dtype), h, tf_cell(2), dtype), [[KAT_2, KAT_2]], atol=TF_TOL)
    np.testing.assert_allclose(cell(np.ones((1, 3), dtype), h, tf_cell(3), dtype), [[KAT_3, KAT_3]], atol=TF_TOL)


def test_oracle_stack_reproduces_tf_testMultiRNNCell():
    w = dict(layers=[tf_cell(2), tf_cell(2)], Wfc=np.eye(2, 6, dtype=np.float32), bfc=np.zeros(6, np.float32))
    st0 = np.full((2, 1, 2), 0.1, np.float32)
    for fwd in (G.gru_forward, G.gru_forward_split):
        logits, st = fwd(w, np.ones((1, 1, 2), np.float32), st0, dtype=np.float64)
        np.testing.assert_allclose(st[:, 0, :].ravel(), [KAT_2, KAT_2, KAT_L1, KAT_L1], atol=TF_TOL)
        np.testing.assert_allclose(logits[0, 0, :2], [KAT_L1, KAT_L1], atol=TF_TOL)   # top layer's output
    lt, _, stt = TE.gru_forward(TE.to_torch(w, torch.float64), torch.ones(1, 1, 2, dtype=torch.float64),
                                torch.from_numpy(st0).double())
    np.testing.assert_allclose(stt.numpy()[:, 0, :].ravel(), [KAT_2, KAT_2, KAT_L1, KAT_L1], atol=TF_TOL)


@pytest.mark.parametrize("n_mel,hidden,n_in,layers", [(40, 128, 2, 2), (40, 128, 3, 1), (60, 64, 3, 2)])
def test_embedding_is_transparent_and_c_oracle_agrees(oracle_c, n_mel, hidden, n_in, layers):
    """The embedding used on the GPU keeps the dead units at exactly 0 and reproduces the constants (numpy + C)."""
    w = embedded(n_mel, hidden, n_in, layers)
    mel, st0 = embedded_inputs(n_mel, hidden, n_in, layers)
    first = KAT_2 if n_in == 2 else KAT_3
    _, st = G.gru_forward(w, mel, st0, dtype=np.float64)
    assert not st[:, :, 2:].any()
    np.testing.assert_allclose(st[0, 0, :2], first, atol=TF_TOL)
    _, _, st_c = oracle_c.gru_forward((n_mel, hidden, layers, 6, 0, -1.0), G.weights_to_blob(w), mel, st0)
    assert not st_c[:, :, 2:].any()
    np.testing.assert_allclose(st_c[0, 0, :2], first, atol=TF_TOL)
    if layers == 2 and n_in == 2:
        np.testing.assert_allclose(st_c[1, 0, :2], KAT_L1, atol=TF_TOL)


# ---------------------------------------------------------------------------------------------- GPU, through the C ABI
def _model(n_mel, hidden, layers, w, kernel, precision="fp32"):
    from keyword_spotting_amd import get_config
    from keyword_spotting_amd.rnn_ctc import DeployModel
    cfg = get_config(n_mel=n_mel, hidden_size=hidden, num_layers=layers, precision=precision)
    return DeployModel(cfg, w, kernel=kernel)


@pytest.mark.gpu
@pytest.mark.parametrize("n_mel,hidden,layers,kernel", [
    (40, 128, 2, "resident"), (40, 128, 2, "generic"), (40, 128, 1, "resident"), (60, 128, 2, "resident"),
    (60, 64, 2, "generic"), (60, 256, 2, "generic")])
@pytest.mark.parametrize("n_in", [2, 3])
@pytest.mark.parametrize("batch", [1, 19])
def test_kws_step_returns_tensorflows_published_constants(n_mel, hidden, layers, kernel, n_in, batch):
    w = embedded(n_mel, hidden, n_in, layers)
    mel, st0 = embedded_inputs(n_mel, hidden, n_in, layers, batch)
    m = _model(n_mel, hidden, layers, w, kernel)
    r = m.forward(torch.from_numpy(mel), torch.from_numpy(st0))
    st, logits = r["state"].cpu().numpy(), r["logits"].cpu().numpy()
    first = KAT_2 if n_in == 2 else KAT_3
    assert not st[:, :, 2:].any()                                  # dead units: exactly 0
    np.testing.assert_allclose(st[0, :, :2], first, atol=TF_TOL)
    top = first
    if layers == 2:
        want_l1 = G.gru_cell(np.full((1, 2), first), np.full((1, 2), 0.1), tf_cell(2), np.float64)[0, 0]
        if n_in == 2:
            assert abs(want_l1 - KAT_L1) < TF_TOL                  # the published MultiRNNCell constant
        np.testing.assert_allclose(st[1, :, :2], want_l1, atol=TF_TOL)
        top = want_l1
    np.testing.assert_allclose(logits[:, 0, :2], top, atol=TF_TOL)


@pytest.mark.gpu
@pytest.mark.parametrize("n_mel,layers", [(40, 2), (40, 1), (60, 3), (32, 2)])
@pytest.mark.parametrize("n_in", [2, 3])
def test_f16x3_split_path_returns_the_published_constants_at_fp32_tolerance(n_mel, layers, n_in):
    """precision="f16x3" (fp16 matrix pipe on split operands) has to meet the fp32 kernels' tolerance, 1e-6 here: 0.5, 1.0 and
    0.1 split exactly or to 2^-22, the rest is the activations' fp32 arithmetic."""
    w = embedded(n_mel, 128, n_in, layers)
    mel, st0 = embedded_inputs(n_mel, 128, n_in, layers, 19)
    m = _model(n_mel, 128, layers, w, "auto", "f16x3")
    r = m.forward(torch.from_numpy(mel), torch.from_numpy(st0))
    st = r["state"].cpu().numpy()
    first = KAT_2 if n_in == 2 else KAT_3
    assert not st[:, :, 2:].any()
    np.testing.assert_allclose(st[0, :, :2], first, atol=TF_TOL)
    x = first
    for l in range(1, layers):
        x = G.gru_cell(np.full((1, 2), x), np.full((1, 2), 0.1), tf_cell(2), np.float64)[0, 0]
        np.testing.assert_allclose(st[l, :, :2], x, atol=TF_TOL)
    if layers >= 2 and n_in == 2:
        np.testing.assert_allclose(st[1, :, :2], KAT_L1, atol=TF_TOL)
    np.testing.assert_allclose(r["logits"].cpu().numpy()[:, 0, :2], x, atol=TF_TOL)


@pytest.mark.gpu
def test_bf16_and_int8_variants_on_the_published_case():
    """0.5 and 1.0 are exact in bf16 / int8; what is left is the rounding of the activations (reported, loose)."""
    w = embedded(40, 128, 2, 2)
    mel, st0 = embedded_inputs(40, 128, 2, 2, 16)
    for precision, tol in (("bf16", 2e-3), ("int8", 2e-2)):
        m = _model(40, 128, 2, w, "auto", precision)
        st = m.forward(torch.from_numpy(mel), torch.from_numpy(st0))["state"].cpu().numpy()
        np.testing.assert_allclose(st[0, :, :2], KAT_2, atol=tol)
        np.testing.assert_allclose(st[1, :, :2], KAT_L1, atol=tol)
