"""Oracle (oracle/decode_oracle.py) pinned against golden vectors generated from the reference's
own utils/prediction.py, utils/queue.py, utils/basic_vad.py (tests/golden/make_decode_golden.py)."""
import numpy as np

from oracle import decode_oracle as D

C = 6


def test_decoders_match_reference_goldens(golden):
    n = int(golden["n_cases"])
    assert n >= 40
    for i in range(n):
        sm = golden["c%d_softmax" % i]
        np.testing.assert_array_equal(D.ctc_decode2(sm, C), golden["c%d_decode2" % i], err_msg="case %d" % i)
        np.testing.assert_array_equal(D.ctc_decode(sm), golden["c%d_decode" % i], err_msg="case %d" % i)
        np.testing.assert_array_equal(D.ctc_decode_strict(sm, C), golden["c%d_strict" % i], err_msg="case %d" % i)
        np.testing.assert_array_equal(D.ctc_decode2(sm, C, thres=0.3), golden["c%d_decode2_t03" % i])
        np.testing.assert_array_equal(D.ctc_decode(sm, lockout=5, thres=0.45, loose_thres=0.25),
                                      golden["c%d_decode_l5" % i])
        np.testing.assert_array_equal(D.ctc_decode_strict(sm, C, lockout=2, thres=0.6),
                                      golden["c%d_strict_l2" % i])
        d2, d1, ds = golden["c%d_decode2" % i], golden["c%d_decode" % i], golden["c%d_strict" % i]
        got = [D.ctc_predict(d2), D.ctc_predict(d1), D.ctc_predict(ds), D.ctc_predict(d2, "12"),
               D.ctc_predict(d1, "33")]
        np.testing.assert_array_equal(got, golden["c%d_predict" % i])


def test_outputs_are_int32_zero_interleaved(golden):
    out = D.ctc_decode2(golden["c5_softmax"], C)
    assert out.dtype == np.int32 and out[0] == 0 and (out[::2] == 0).all() and (out[1::2] > 0).all()


def test_empty_window():
    for f in (lambda s: D.ctc_decode2(s, C), D.ctc_decode, lambda s: D.ctc_decode_strict(s, C)):
        np.testing.assert_array_equal(f(np.zeros((0, C), np.float32)), [0])


def test_ctc_predict_sequences(golden):
    for i in range(int(golden["n_pseq"])):
        s = golden["p%d_seq" % i]
        got = [D.ctc_predict(s), D.ctc_predict(s, "123"), D.ctc_predict(s, "33")]
        np.testing.assert_array_equal(got, golden["p%d_out" % i])


def test_evaluate(golden):
    got = D.evaluate(golden["eval_result"].tolist(), golden["eval_target"].tolist())
    np.testing.assert_array_equal(got, golden["eval_out"])


def test_simple_queue_trace(golden):
    q = D.SimpleQueue(15)
    for k, op in enumerate(golden["q_ops"]):
        q.add(k) if op == 0 else q.clear()
        content = q.get_all()
        assert q.len == golden["q_len"][k]
        assert int(q.full()) == golden["q_full"][k]
        assert len(content) == golden["q_n"][k]
        assert (content[0] if content else -1) == golden["q_head"][k]


def test_vad(golden):
    sig = golden["vad_sig"]
    np.testing.assert_array_equal([int(D.vad(s, 30)) for s in sig], golden["vad_30"])
    np.testing.assert_array_equal([int(D.vad(s)) for s in sig], golden["vad_40"])


def test_chunk_frame_counts():
    # 3600-sample hops -> 21 frames first, then 22/23 alternating (22.5 average)
    counts = D.chunk_frame_counts([3600] * 9)
    assert counts == [21, 22, 23, 22, 23, 22, 23, 22, 23]   # hand-derived from detector.py:181-183
    # the carry makes framing seamless: chunked frame total == one-shot framing of the whole signal
    for sizes in ([3600] * 20, [1234, 4000, 800, 160, 159, 4001], [400], [399, 1]):
        total = sum(sizes)
        assert sum(D.chunk_frame_counts(sizes)) == D.frames_in(total), sizes
    assert D.carry_len(3600) == 240 + (3600 - 400) % 160
