"""HIP decode / predict / vad kernels against the reference goldens (bit-exact integer outputs)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
C = 6


def test_decoders_match_reference_goldens(golden):
    from keyword_spotting_amd import prediction as P
    for i in range(int(golden["n_cases"])):
        sm = golden["c%d_softmax" % i]
        np.testing.assert_array_equal(P.ctc_decode2(sm, C), golden["c%d_decode2" % i], err_msg="case %d" % i)
        np.testing.assert_array_equal(P.ctc_decode(sm), golden["c%d_decode" % i], err_msg="case %d" % i)
        np.testing.assert_array_equal(P.ctc_decode_strict(sm, C), golden["c%d_strict" % i], err_msg="case %d" % i)
        np.testing.assert_array_equal(P.ctc_decode2(sm, C, thres=0.3), golden["c%d_decode2_t03" % i])
        np.testing.assert_array_equal(P.ctc_decode(sm, lockout=5, thres=0.45, loose_thres=0.25),
                                      golden["c%d_decode_l5" % i])
        np.testing.assert_array_equal(P.ctc_decode_strict(sm, C, lockout=2, thres=0.6), golden["c%d_strict_l2" % i])
        d2, d1, ds = golden["c%d_decode2" % i], golden["c%d_decode" % i], golden["c%d_strict" % i]
        got = [P.ctc_predict(d2), P.ctc_predict(d1), P.ctc_predict(ds), P.ctc_predict(d2, "12"), P.ctc_predict(d1, "33")]
        np.testing.assert_array_equal(got, golden["c%d_predict" % i])


def test_batched_ragged_decode_and_predict(golden):
    """All golden cases in ONE launch: ragged lengths padded to the longest window."""
    from keyword_spotting_amd import _lib, prediction as P
    n = int(golden["n_cases"])
    sms = [golden["c%d_softmax" % i] for i in range(n)]
    tmax = max(s.shape[0] for s in sms)
    batch = np.full((n, tmax, C), 0.99, np.float32)       # poison beyond each length
    for i, s in enumerate(sms):
        batch[i, :s.shape[0]] = s
    lens = np.array([s.shape[0] for s in sms], np.int32)
    for kind, key in ((_lib.DECODE2, "decode2"), (_lib.DECODE, "decode"), (_lib.DECODE_STRICT, "strict")):
        thres = 0.4 if kind == _lib.DECODE2 else 0.5
        words, counts = P.decode_batch(kind, batch, lens, 3, thres, 0.2)
        hits = P.ctc_predict((words, counts), "1233").cpu().numpy()
        w, c = words.cpu().numpy(), counts.cpu().numpy()
        for i in range(n):
            want = golden["c%d_%s" % (i, key)]
            np.testing.assert_array_equal(w[i, :c[i]], want[1::2], err_msg="%s case %d" % (key, i))
            assert hits[i] == golden["c%d_predict" % i][{"decode2": 0, "decode": 1, "strict": 2}[key]]


def test_truncation_reports_full_count(golden):
    from keyword_spotting_amd import _lib, prediction as P
    sm = golden["c15_softmax"]
    full = golden["c15_decode2"][1::2]
    words, counts = P.decode_batch(_lib.DECODE2, sm, None, 3, 0.4, 0.0, max_words=3)
    assert int(counts[0]) == len(full) and len(full) > 3
    np.testing.assert_array_equal(words.cpu().numpy()[0], full[:3])


def test_ctc_predict_sequences(golden):
    from keyword_spotting_amd import prediction as P
    for i in range(int(golden["n_pseq"])):
        s = golden["p%d_seq" % i]
        got = [P.ctc_predict(s), P.ctc_predict(s, "123"), P.ctc_predict(s, "33")]
        np.testing.assert_array_equal(got, golden["p%d_out" % i])


def test_decode_argument_errors():
    from keyword_spotting_amd import _lib, prediction as P
    with pytest.raises(_lib.InvalidArgumentError):
        P.ctc_decode(np.zeros((4, 4), np.float32))          # ctc_decode needs >= 5 columns
    with pytest.raises(_lib.InvalidArgumentError):
        P.ctc_decode2(np.zeros((4, 6), np.float32), 5)      # classnum mismatch
    with pytest.raises(_lib.InvalidArgumentError):
        P.ctc_predict([1, 2], "12a")


def test_vad(golden):
    from keyword_spotting_amd.basic_vad import vad
    sig = golden["vad_sig"]
    speech30, sums = vad(torch.from_numpy(sig), 30, return_sum=True)
    np.testing.assert_allclose(sums.cpu().numpy(), golden["vad_sum"], rtol=2e-6)
    clear = np.abs(golden["vad_sum"] - 30) > 1e-3            # fp32 summation order differs from numpy's
    np.testing.assert_array_equal(speech30.cpu().numpy()[clear], golden["vad_30"][clear])
    clear = np.abs(golden["vad_sum"] - 40) > 1e-3
    np.testing.assert_array_equal(vad(torch.from_numpy(sig)).cpu().numpy()[clear], golden["vad_40"][clear])
    assert vad(sig[5], 30) == bool(golden["vad_30"][5])
    assert vad(np.zeros(0, np.float32)) is False


def test_simple_queue_trace(golden):
    from keyword_spotting_amd.queue import SimpleQueue
    q = SimpleQueue(15)
    for k, op in enumerate(golden["q_ops"]):
        q.add(k) if op == 0 else q.clear()
        content = q.get_all()
        assert (q.len, int(q.full()), len(content), content[0] if content else -1) == \
            (golden["q_len"][k], golden["q_full"][k], golden["q_n"][k], golden["q_head"][k])
