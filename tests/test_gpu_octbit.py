"""HIP OctbitMatMul: the reference's known-answer tests + bit-exact agreement with the C oracle."""
import numpy as np
import pytest
import torch

from oracle import octbit_oracle as O

pytestmark = pytest.mark.gpu


def test_reference_known_answers():
    """octbit/octbit_ops_test.py:24-34 and :41-53, literal values."""
    from keyword_spotting_amd.octbit_ops import octbit_mat_mul
    x1 = [[-1.0 for _ in range(64)]]
    x2 = np.array([[i for i in range(64)]], np.int8)
    r = octbit_mat_mul(x1, x2, scale=3.0, bias=[127 * 2016.0])
    np.testing.assert_array_equal(r.cpu().numpy(), [[-6048.0]])
    x1 = [[-1 for _ in range(64)] for _ in range(2)]
    x2 = np.array([[1 for _ in range(64)]] + [[i for i in range(64)] for _ in range(3)], np.int8)
    r = octbit_mat_mul(x1, x2, scale=2.0, bias=[127 * 64.0, 127 * 2016.0, 127 * 2016.0, 127 * 2016.0])
    np.testing.assert_array_equal(r.cpu().numpy(), [[-128.0, -4032.0, -4032.0, -4032.0]] * 2)


def test_bit_exact_vs_oracle(oracle_c):
    from keyword_spotting_amd.octbit_ops import octbit_mat_mul
    rng = np.random.default_rng(81)
    for trial, (a, k, n) in enumerate([(1, 256, 256), (1, 256, 128), (7, 128, 6), (3, 64, 300), (2, 512, 33)]):
        x = rng.standard_normal((a, k)).astype(np.float32) * 2
        if trial == 1:
            x = np.abs(x)
        if trial == 3:
            x[:] = 5.0
            x[0, 0] = -5.0
        wq = rng.integers(-127, 128, (n, k)).astype(np.int8)
        if trial == 3:
            wq[:] = 127                                       # saturating pairs
        bias = (127.0 * wq.astype(np.float64).sum(1)).astype(np.float32)
        rc, want = oracle_c.octbit_matmul(x, wq, 0.0123, bias)
        assert rc == 0
        got = octbit_mat_mul(x, wq, scale=0.0123, bias=bias).cpu().numpy()
        np.testing.assert_array_equal(got, want, err_msg="trial %d" % trial)


def test_per_row_scale_equals_row_by_row(oracle_c):
    """A batch of independent streams must be quantised per stream to reproduce the batch-1 op."""
    from keyword_spotting_amd.octbit_ops import octbit_mat_mul
    rng = np.random.default_rng(82)
    x = (rng.standard_normal((33, 256)) * rng.uniform(0.1, 5, (33, 1))).astype(np.float32)
    x[4] = np.abs(x[4])
    wq = rng.integers(-127, 128, (128, 256)).astype(np.int8)
    bias = (127.0 * wq.astype(np.float64).sum(1)).astype(np.float32)
    got = octbit_mat_mul(x, wq, scale=0.02, bias=bias, per_row_scale=True).cpu().numpy()
    for a in range(33):
        _, want = oracle_c.octbit_matmul(x[a:a + 1], wq, 0.02, bias)
        np.testing.assert_array_equal(got[a:a + 1], want)


def test_preconditions_raise():
    from keyword_spotting_amd import _lib
    from keyword_spotting_amd.octbit_ops import octbit_mat_mul
    x, w = np.ones((1, 64), np.float32), np.ones((2, 64), np.int8)
    with pytest.raises(_lib.InvalidArgumentError, match="positive"):
        octbit_mat_mul(x, w, scale=0.0, bias=[0, 0])
    with pytest.raises(_lib.InvalidArgumentError):
        octbit_mat_mul(x[:, :32], w[:, :32], scale=1.0, bias=[0, 0])
    with pytest.raises(_lib.InvalidArgumentError, match="transposed"):
        octbit_mat_mul(x, w, transpose_b=False, scale=1.0, bias=[0, 0])
    with pytest.raises(_lib.InvalidArgumentError, match="f is not equal"):
        octbit_mat_mul(x, np.ones((2, 128), np.int8), scale=1.0, bias=[0, 0])


def test_quantiser_matches_oracle():
    from keyword_spotting_amd.octbit_graph import default_octbit_matmul_name_check, octize_weight_int8_signed
    rng = np.random.default_rng(83)
    for shape in ((256, 256), (256, 128), (128, 6)):
        w = (rng.standard_normal(shape) * 0.3).astype(np.float32)
        wq, scale, bias = octize_weight_int8_signed(w)
        wq2, scale2, bias2 = O.octize_weight_int8_signed(w)
        np.testing.assert_array_equal(wq, wq2)
        assert scale == np.float32(scale2)
        np.testing.assert_array_equal(bias, bias2.astype(np.float32))
    assert default_octbit_matmul_name_check("model/drnn/cell_1/gates/MatMul")
    assert not default_octbit_matmul_name_check("model/drnn/cell_0/gates/MatMul")
