"""The incremental decode window on the device (csrc/window_device.h): as a launch of its own (kws_window_step_incremental)
and as the tail of the last GRU layer's launch (kws_stream_feed) -- against the re-scanning kws_window_step, the algorithm's
Python statement (tests/window_model.py), the oracle's SimpleQueue + ctc_decode2 + ctc_predict replay, and traces generated
by the reference's own modules (tests/golden/window_golden.npz).  Reference: detector.py:168-177,195-209, utils/queue.py:26-32,
utils/prediction.py:65-86,111-118."""
import ctypes
import os

import numpy as np
import pytest
import torch

from oracle import decode_oracle as D
from oracle import gru_oracle as G
from tests.window_model import IncrementalWindow

pytestmark = pytest.mark.gpu
C = 6
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "window_golden.npz")


def rows_for(words):
    """[B,T] words (-1..3) -> softmax [B,T,6] whose ctc_decode2 frame word is that word."""
    words = np.asarray(words)
    sm = np.full(words.shape + (C,), 0.02, np.float32)
    idx = np.where(words < 0, C - 1, words + 1)
    np.put_along_axis(sm, idx[..., None], 0.9, axis=-1)
    return sm


class DeviceWindow(object):
    def __init__(self, b, nq, tmax, incremental, thres=0.4, classes=C):
        from keyword_spotting_amd import _lib
        self._lib, self.lib, self.b = _lib, _lib.load(), b
        self.h = ctypes.c_void_p()
        _lib.check(self.lib.kws_window_create(b, nq, tmax, classes, thres, ctypes.byref(self.h)))
        self.fn = self.lib.kws_window_step_incremental if incremental else self.lib.kws_window_step
        self.hit = torch.zeros(b, dtype=torch.int32, device="cuda")
        self.restart = torch.zeros(b, dtype=torch.uint8, device="cuda")

    def step(self, sm, clear, label):
        sm_d = torch.from_numpy(np.ascontiguousarray(sm)).cuda()
        cl = torch.from_numpy(np.asarray(clear, np.uint8)).cuda()
        self._lib.check(self.fn(self.h, self._lib.ptr(sm_d) if sm.shape[1] else None, int(sm.shape[1]), self._lib.ptr(cl), label.encode(),
                                self._lib.ptr(self.hit), self._lib.ptr(self.restart), self._lib.current_stream_ptr()))
        hit = self.hit.cpu().numpy()
        assert np.array_equal(self.restart.cpu().numpy(), hit.astype(np.uint8))
        return hit

    def close(self):
        self.lib.kws_window_destroy(self.h)


@pytest.mark.parametrize("nq,tmax,label", [(15, 23, "1233"), (1, 9, "3"), (2, 40, "12"), (3, 23, "121"), (17, 5, "33"), (24, 16, "1212"), (32, 16, "1212"),
                                           (64, 7, "2312"), (15, 23, "123412341234123")])
def test_incremental_kernel_equals_the_rescan_kernel_the_model_and_the_oracle(nq, tmax, label):
    """Random word plateaus over many chunks of ragged length (0 frames included), per stream: windows fill, evict with a word
    held across the evicted chunk, clear on silence and restart on hits.  Four implementations must agree on every chunk."""
    rng = np.random.default_rng(7000 + nq)
    b, steps = 37, 140                                      # 37: a ragged last group of 5 streams
    inc, ref = DeviceWindow(b, nq, tmax, True), DeviceWindow(b, nq, tmax, False)
    model = [IncrementalWindow(nq, [int(c) for c in label]) for _ in range(b)]
    queues = [D.SimpleQueue(nq) for _ in range(6)]          # the oracle replay for the first streams (it re-scans: slow)
    word = rng.integers(-1, 4, b)
    total = 0
    for step in range(steps):
        t = int(rng.choice([0, 1, 2, tmax // 2, tmax - 1, tmax]))
        words = np.zeros((b, t), np.int64)
        for k in range(t):
            change = rng.random(b) < 0.15
            word = np.where(change, rng.integers(-1, 4, b), word)
            words[:, k] = word
        clear = (rng.random(b) < 0.02).astype(np.uint8)
        sm = rows_for(words)
        got_inc, got_ref = inc.step(sm, clear, label), ref.step(sm, clear, label)
        np.testing.assert_array_equal(got_inc, got_ref, err_msg="step %d" % step)
        want = np.asarray([model[s].step(words[s].tolist(), bool(clear[s])) for s in range(b)])
        np.testing.assert_array_equal(got_inc, want, err_msg="step %d (model)" % step)
        for s, q in enumerate(queues):
            if clear[s]:
                q.clear()
            q.add(sm[s])
            o = D.ctc_predict(D.ctc_decode2(np.concatenate(q.get_all(), 0), C, 0.4), label)
            assert int(got_inc[s]) == int(o), (step, s)
            if o:
                q.clear()
        total += int(got_inc.sum())
    assert total > 0 or len(label) > 4
    inc.close()
    ref.close()


def test_incremental_kernel_property_against_the_rescan_oracle():
    """hypothesis drives the DEVICE kernel: arbitrary chunk lists (empty chunks, held words straddling chunk and eviction
    boundaries, clears), small and reference-sized windows, several labels -- every decision equals the oracle's
    SimpleQueue + concatenate + ctc_decode2 + ctc_predict replay (the same strategies as tests/test_window_incremental.py)."""
    from hypothesis import HealthCheck, given, settings
    from hypothesis import strategies as st
    from tests.test_window_incremental import RescanWindow
    chunk = st.lists(st.integers(-1, 3), min_size=0, max_size=7)
    held = st.builds(lambda w, n, tail: [w] * n + tail, st.integers(-1, 3), st.integers(0, 6), st.lists(st.integers(-1, 3), max_size=2))

    @settings(max_examples=120, deadline=None, suppress_health_check=[HealthCheck.too_slow])
    @given(chunks=st.lists(st.tuples(st.one_of(chunk, held), st.booleans()), min_size=1, max_size=30),
           nq=st.sampled_from([1, 2, 3, 4, 15]), label=st.sampled_from(["1233", "12", "1", "11", "121", "33", "1212"]))
    def run(chunks, nq, label):
        win, ref = DeviceWindow(3, nq, 9, True), RescanWindow(nq, label)
        try:
            for k, (words, flag) in enumerate(chunks):
                clear = bool(flag and k % 3 == 0)
                w = np.asarray(words, np.int64).reshape(1, -1)
                got = win.step(rows_for(np.repeat(w, 3, 0)), [clear] * 3, label)
                want = ref.step(words, clear)
                assert (got == want).all(), (k, words, clear, nq, label, got, want)
        finally:
            win.close()
    run()


def test_incremental_kernel_on_the_reference_generated_traces():
    g = np.load(GOLDEN)
    hits = 0
    for name in g["names"]:
        words, lens, silent = g[name + "_words"], g[name + "_lens"], g[name + "_silent"]
        label = "".join(str(int(d)) for d in g[name + "_label"])
        nq = int(g[name + "_maxlen"])
        b = 3                                               # the same trace on three streams (one of them with a silent first chunk)
        win = DeviceWindow(b, nq, max(int(lens.max()), 1), True)
        pos = 0
        for k, n in enumerate(lens):
            w = words[pos:pos + n].astype(np.int64)
            pos += n
            sm = rows_for(np.broadcast_to(w, (b, n)))
            got = win.step(sm, [silent[k]] * b, label)
            assert (got == int(g[name + "_hit"][k])).all(), (name, k, got)
            hits += int(got[0])
        win.close()
    assert hits > 20


def test_frame_rule_and_threshold_of_the_incremental_kernel():
    """Words come from the same frame rule as ctc_decode2 (first maximum over classes 1..C-2, strictly above the threshold):
    seeded Dirichlet rows, thresholds 0.4 and 0.3, C = 6 and C = 8, against the oracle."""
    rng = np.random.default_rng(7100)
    for classes, thres, label in ((6, 0.4, "12"), (6, 0.3, "31"), (8, 0.4, "56"), (4, 0.4, "21")):
        b, nq = 20, 4
        win = DeviceWindow(b, nq, 12, True, thres=thres, classes=classes)
        queues = [D.SimpleQueue(nq) for _ in range(b)]
        fired = 0
        for step in range(60):
            t = int(rng.integers(0, 13))
            sm = rng.dirichlet([0.25] * classes, (b, t)).astype(np.float32) if t else np.zeros((b, 0, classes), np.float32)
            got = win.step(sm, np.zeros(b, np.uint8), label)
            for s in range(b):
                queues[s].add(sm[s])
                o = D.ctc_predict(D.ctc_decode2(np.concatenate(queues[s].get_all(), 0), classes, thres), label)
                assert int(got[s]) == int(o), (classes, step, s)
                if o:
                    queues[s].clear()
                    fired += 1
        assert fired > 0
        win.close()


def test_the_label_is_bound_to_the_incremental_state():
    from keyword_spotting_amd import _lib
    win = DeviceWindow(4, 15, 8, True)
    sm = rows_for(np.zeros((4, 3), np.int64))
    win.step(sm, np.zeros(4, np.uint8), "12")
    win.step(sm, np.zeros(4, np.uint8), "12")
    with pytest.raises(_lib.InvalidArgumentError):
        win.step(sm, np.zeros(4, np.uint8), "13")           # the queued summaries were built for '12'
    with pytest.raises(_lib.InvalidArgumentError):
        win.step(sm, np.zeros(4, np.uint8), "1x")
    with pytest.raises(_lib.InvalidArgumentError):
        DeviceWindow(4, 15, 8, True).step(sm, np.zeros(4, np.uint8), "1234123412341234")      # 16 digits: the matcher has 16 states
    assert win.step(sm, np.zeros(4, np.uint8), "12").tolist() == [0, 0, 0, 0]
    # argument checks mirror kws_window_step's: T outside [0, max_frames], null pointers
    lib = _lib.load()
    assert lib.kws_window_step_incremental(win.h, _lib.ptr(win.hit), 9, None, b"12", _lib.ptr(win.hit), None, None) == _lib.KWS_ERR_INVALID_ARGUMENT
    assert lib.kws_window_step_incremental(win.h, None, 3, None, b"12", _lib.ptr(win.hit), None, None) == _lib.KWS_ERR_INVALID_ARGUMENT
    assert lib.kws_window_step_incremental(win.h, None, 0, None, b"12", None, None, None) == _lib.KWS_ERR_INVALID_ARGUMENT
    assert lib.kws_window_step_incremental(None, None, 0, None, b"12", _lib.ptr(win.hit), None, None) == _lib.KWS_ERR_INVALID_ARGUMENT
    win.close()
    empty = DeviceWindow(2, 3, 4, True)
    assert empty.step(sm[:2], np.zeros(2, np.uint8), "").tolist() == [1, 1]     # '' occurs in anything (utils/prediction.py:118)
    empty.close()


def _keyword_weights(seed=0):
    w = G.init_weights(seed=seed)
    w["Wfc"] = (w["Wfc"] * 3.0).astype(np.float32)
    return w


@pytest.mark.parametrize("precision,kernel,layers,fused", [("fp32", "auto", 2, True), ("f16x3", "auto", 2, True), ("bf16", "auto", 2, True),
                                                           ("f16x3", "auto", 1, True), ("f16x3", "auto", 3, True),
                                                           ("fp32", "generic", 2, False), ("int8", "auto", 2, False), ("fp32", "auto", 1, False),
                                                           ("bf16", "auto", 1, False)])
def test_stream_feed_carries_the_window_in_the_last_layer_launch(precision, kernel, layers, fused):
    """kws_stream_feed: where the last layer's kernel has the window tail the chunk is three launches (two for bf16) and the
    launch name says so; everywhere else window_inc_kernel follows.  Either way the decisions are the host mirror's
    (HotwordDetector: SimpleQueue + windowed ctc_decode2 + ctc_predict on the same kernels' softmax), chunk by chunk, with
    silence clears, sub-frame chunks, evictions and trigger restarts -- and the carried state is identical."""
    from keyword_spotting_amd import get_config
    from keyword_spotting_amd.detector import HotwordDetector, StreamManager
    from keyword_spotting_amd.frontend import MelFrontend
    from keyword_spotting_amd.rnn_ctc import DeployModel
    cfg = get_config(precision=precision, num_layers=layers)
    b = 41
    rng = np.random.default_rng(7201)
    # weights and labels this random model really emits on noise (random weights never spell "1233"): the first seed whose
    # model says anything, its most frequent word and bigram
    fe = MelFrontend(cfg)
    noise = torch.from_numpy((rng.standard_normal((b, 16000)) * 0.2).astype(np.float32))
    for seed in range(7200, 7240):
        w = G.random_weights(40, 128, layers, 6, seed=seed)
        w["Wfc"] = (w["Wfc"] * 4.0).astype(np.float32)
        probe = DeployModel(cfg, w, kernel=kernel)
        sm = probe.forward(fe.forward(noise), probe.zero_state(b), want_logits=False)["softmax"].cpu().numpy()
        probe.close()
        seqs = [D.ctc_decode2(sm[k], 6)[1::2] for k in range(b)]
        words = np.concatenate(seqs)
        if words.size >= 2 * b:
            break
    assert words.size >= 2 * b
    labels = [str(int(np.bincount(words).argmax()))]
    pairs = [10 * int(q[i]) + int(q[i + 1]) for q in seqs for i in range(len(q) - 1)]
    if pairs:
        labels.append(str(int(np.bincount(pairs).argmax())))
    for label in labels:
        md, mm = DeployModel(cfg, w, kernel=kernel), DeployModel(cfg, w, kernel=kernel)
        det = HotwordDetector(md, batch=b, label=label, window_chunks=4)
        mgr = StreamManager(mm, batch=b, label=label, window_chunks=4)
        total, carry = 0, 0
        for ci in range(40):
            n = int(rng.choice([3600, 3600, 3600, 1800, 200, 5000])) if ci else 200      # the first chunk is shorter than a frame
            pcm = (rng.standard_normal((b, n)) * 0.2).astype(np.float32)
            quiet = rng.random(b) < 0.06
            pcm[quiet] *= 1e-4                                   # below the VAD threshold: state reset + window clear
            x = torch.from_numpy(pcm).cuda()
            want = np.zeros(b, np.int32)
            want[det.feed_pcm(x, fe)] = 1
            got = mgr.feed_pcm(x, fe).cpu().numpy()
            np.testing.assert_array_equal(got, want, err_msg="%s label %s chunk %d" % (precision, label, ci))
            assert torch.equal(mgr.state, det.state), (precision, ci)
            total += int(want.sum())
            frames = D.frames_in(carry + n)
            carry = D.carry_len(carry + n) if frames else carry + n
            if frames:                                           # (a zero-frame chunk launches no GRU kernel: window_inc_kernel steps the window)
                names = mm.kernel_names()
                assert any("window tail" in nm for nm in names) == fused, (names, n)
        assert total > 0
        mgr.close()
        md.close()
        mm.close()


def test_where_the_window_rides_and_where_it_follows():
    """The tail rides in the last layer's launch when every workgroup takes ONE group of streams and the window is at most 24
    chunks (what it stages in LDS); persistent workgroups (B > 16 x CUs) and longer windows are followed by window_inc_kernel
    instead.  Decisions equal the mel-fed manager's either way, ragged last group included."""
    from keyword_spotting_amd import get_config
    from keyword_spotting_amd.detector import StreamManager
    from keyword_spotting_amd.frontend import MelFrontend
    from keyword_spotting_amd.rnn_ctc import DeployModel
    cfg = get_config(precision="bf16")
    w = _keyword_weights(seed=9)
    rng = np.random.default_rng(7300)
    fe = MelFrontend(cfg)
    probe = DeployModel(cfg, w)
    sm = probe.forward(fe.forward(torch.from_numpy((rng.standard_normal((64, 16000)) * 0.2).astype(np.float32))),
                       probe.zero_state(64), want_logits=False)["softmax"].cpu().numpy()
    probe.close()
    words = np.concatenate([D.ctc_decode2(sm[k], 6)[1::2] for k in range(64)])
    assert words.size > 0
    label = str(int(np.bincount(words).argmax()))
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    for b, nq, fused in ((16 * (cus + 40) + 5, 15, False), (16 * (cus - 50) + 5, 24, True), (16 * (cus - 50) + 5, 25, False)):
        ma, mb = DeployModel(cfg, w), DeployModel(cfg, w)
        a = StreamManager(ma, batch=b, label=label, window_chunks=nq)
        ref = StreamManager(mb, batch=b, label=label, window_chunks=nq)
        carry = torch.zeros(b, 0, device="cuda")
        total = 0
        for ci in range(6):
            x = torch.from_numpy((rng.standard_normal((b, 3600)) * 0.2).astype(np.float32)).cuda()
            got = a.feed_pcm(x, fe).cpu().numpy()
            assert any("window tail" in nm for nm in ma.kernel_names()) == fused, (b, nq)
            data = torch.cat([carry, x], 1)
            keep = (data.shape[1] - 400) % 160 + 240
            carry = data[:, data.shape[1] - keep:].contiguous()
            want = ref.feed(fe.forward(data.contiguous()), pcm_chunk=x).cpu().numpy()
            np.testing.assert_array_equal(got, want, err_msg="b %d nq %d chunk %d" % (b, nq, ci))
            total += int(want.sum())
        assert total > 0
        a.close(); ref.close(); ma.close(); mb.close()


def _emitting_model(cfg, fe, b, rng, kernel="auto", layers=2):
    """Random weights whose model really says something on noise (random weights never spell "1233"): the first seed that emits
    at least two words per stream; -> (weights, its most frequent word as a one-digit label, the decoded sequences)."""
    from keyword_spotting_amd.rnn_ctc import DeployModel
    noise = torch.from_numpy((rng.standard_normal((b, 16000)) * 0.2).astype(np.float32))
    for seed in range(7200, 7260):
        w = G.random_weights(40, 128, layers, 6, seed=seed)
        w["Wfc"] = (w["Wfc"] * 4.0).astype(np.float32)
        probe = DeployModel(cfg, w, kernel=kernel)
        sm = probe.forward(fe.forward(noise), probe.zero_state(b), want_logits=False)["softmax"].cpu().numpy()
        probe.close()
        seqs = [D.ctc_decode2(sm[k], 6)[1::2] for k in range(b)]
        words = np.concatenate(seqs)
        if words.size >= 2 * b:
            return w, str(int(np.bincount(words).argmax())), seqs
    raise AssertionError("no seed gives a model that emits words")


@pytest.mark.parametrize("precision", ["fp32", "f16x3", "bf16"])
def test_long_chunks_at_the_tails_frame_limit(precision):
    """The fused tail keeps a call's frame words in LDS for up to 64 frames: chunks that give exactly 63 / 64 frames ride, 65 and
    more are followed by window_inc_kernel; decisions equal the host mirror's either way (several 16-frame flush blocks per call)."""
    from keyword_spotting_amd import get_config
    from keyword_spotting_amd.detector import HotwordDetector, StreamManager
    from keyword_spotting_amd.frontend import MelFrontend
    from keyword_spotting_amd.rnn_ctc import DeployModel
    cfg = get_config(precision=precision)
    fe = MelFrontend(cfg)
    b = 19
    rng = np.random.default_rng(7400)
    w, label, _ = _emitting_model(cfg, fe, b, rng)
    md, mm = DeployModel(cfg, w), DeployModel(cfg, w)
    det = HotwordDetector(md, batch=b, label=label, window_chunks=5)
    mgr = StreamManager(mm, batch=b, label=label, window_chunks=5, max_frames=80)
    carry, seen, total = 0, set(), 0
    for n in (400 + 160 * 62, 160 * 64 - 0, 160 * 63 + 7, 160 * 65, 160 * 64, 160 * 70, 3600, 160 * 63):
        x = torch.from_numpy((rng.standard_normal((b, n)) * 0.2).astype(np.float32)).cuda()
        want = np.zeros(b, np.int32)
        want[det.feed_pcm(x, fe)] = 1
        got = mgr.feed_pcm(x, fe).cpu().numpy()
        frames = D.frames_in(carry + n)
        carry = D.carry_len(carry + n)
        np.testing.assert_array_equal(got, want, err_msg="%s chunk of %d samples (%d frames)" % (precision, n, frames))
        assert torch.equal(mgr.state, det.state)
        # (fp32 below half a chip of streams runs calls of >= 64 frames with its layers overlapped on HIP streams: no tail there)
        rides = frames <= 64 and not (precision == "fp32" and frames >= 64)
        assert any("window tail" in nm for nm in mm.kernel_names()) == rides, (frames, mm.kernel_names())
        seen.add(frames)
        total += int(want.sum())
    assert {63, 64}.intersection(seen) and any(f > 64 for f in seen) and total > 0, (seen, total)
    mgr.close(); md.close(); mm.close()
