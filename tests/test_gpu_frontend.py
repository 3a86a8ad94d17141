"""PCM -> mel front-end (models/rnn_ctc.py:134-149) against the numpy restatement (PARITY UNPINNED: librosa and
TF are absent; see oracle/frontend_oracle.py) and its structural properties."""
import numpy as np
import pytest
import torch

from oracle import decode_oracle as D
from oracle import frontend_oracle as F
from oracle import gru_oracle as G

pytestmark = pytest.mark.gpu


def _frontend(**kw):
    from keyword_spotting_amd import get_config
    from keyword_spotting_amd.frontend import MelFrontend
    cfg = get_config(**kw)
    return cfg, MelFrontend(cfg)


@pytest.mark.parametrize("n_mel", [40, 60])
def test_mel_basis_matches_restated_librosa(n_mel):
    cfg, fe = _frontend(n_mel=n_mel)
    got = fe.mel_basis()
    want = F.mel_basis(16000, 400, n_mel, 300.0, 8000.0)
    np.testing.assert_allclose(got, want, rtol=2e-6, atol=1e-9)
    assert got.shape == (n_mel, 201) and (got >= 0).all()
    # Slaney area normalisation: every filter integrates to ~1 Hz^-1 * bin width (2/(f_hi-f_lo) * triangle area)
    peaks = got.argmax(1)
    assert (np.diff(peaks) > 0).all() and peaks[0] >= 7          # 300 Hz = bin 7.5
    assert got[:, :7].sum() == 0.0                               # nothing below fmin


@pytest.mark.parametrize("n_mel,batch,n", [(40, 3, 3600), (60, 2, 4000), (40, 1, 400), (40, 5, 16000), (40, 2, 399)])
def test_melspec_matches_oracle(n_mel, batch, n):
    cfg, fe = _frontend(n_mel=n_mel)
    rng = np.random.default_rng(200 + n)
    pcm = (rng.standard_normal((batch, n)) * 0.1).astype(np.float32)
    got = fe.forward(torch.from_numpy(pcm)).cpu().numpy()
    want = F.melspec(pcm, n_mels=n_mel)
    assert got.shape == want.shape == (batch, D.frames_in(n), n_mel)
    if want.size:
        scale = np.abs(want).max()
        assert np.abs(got - want).max() < 2e-5 * scale, (np.abs(got - want).max(), scale)


def test_pure_tone_lands_in_the_right_filter():
    cfg, fe = _frontend()
    t = np.arange(4000) / 16000.0
    basis = fe.mel_basis()
    for hz in (440.0, 1000.0, 3000.0, 6000.0):
        mel = fe.forward(torch.from_numpy(np.sin(2 * np.pi * hz * t).astype(np.float32)))   # rank-1 input
        assert mel.dim() == 2 and mel.shape[1] == 40
        k = int(round(hz / 40.0))                                 # 40 Hz per bin
        assert int(mel.mean(0).argmax()) == int(basis[:, k].argmax())


def test_chunked_framing_with_carry_equals_one_shot():
    """detector.py:179-183: prepending the carried tail makes the chunked frames exactly the one-shot frames."""
    from keyword_spotting_amd.detector import ChunkFramer
    cfg, fe = _frontend()
    rng = np.random.default_rng(210)
    pcm = (rng.standard_normal((2, 3600 * 7 + 123)) * 0.05).astype(np.float32)
    whole = fe.forward(torch.from_numpy(pcm))
    parts, res, pos, fr = [], pcm[:, :0], 0, ChunkFramer()
    for n in [3600] * 7 + [123]:
        data = np.concatenate([res, pcm[:, pos:pos + n]], 1)
        keep = (data.shape[1] - 400) % 160 + 240
        res = data[:, -keep:]
        m = fe.forward(torch.from_numpy(data.copy()))
        assert m.shape[1] == fr.push(n)
        parts.append(m)
        pos += n
    got = torch.cat(parts, 1)
    assert got.shape == whole.shape
    assert torch.equal(got, whole)                                # same frames, same instruction sequence


def test_detector_feed_pcm_end_to_end():
    from keyword_spotting_amd.detector import HotwordDetector
    from keyword_spotting_amd.rnn_ctc import DeployModel
    cfg, fe = _frontend()
    w = G.init_weights()
    w["Wfc"] = (w["Wfc"] * 3).astype(np.float32)
    model = DeployModel(cfg, w)
    det = HotwordDetector(model, batch=2, label="12")
    rng = np.random.default_rng(211)
    pcm = (rng.standard_normal((2, 3600 * 12)) * 0.2).astype(np.float32)
    fired = []
    for c in range(12):
        fired.append(det.feed_pcm(pcm[:, 3600 * c:3600 * (c + 1)], fe))
    # oracle replay: mel by the oracle front-end on the same chunking, policy as detector.py
    chunks = D.chunk_frame_counts([3600] * 12)
    mel = F.melspec(pcm).astype(np.float32)
    assert mel.shape[1] == sum(chunks)
    for s in range(2):
        state = np.zeros((2, 1, 128), np.float64)
        q, pos, want, ok = D.SimpleQueue(15), 0, [], True
        for ci, n in enumerate(chunks):
            lg, state = G.gru_forward(w, mel[s:s + 1, pos:pos + n], state, dtype=np.float64)
            sm = G.softmax(lg)[0]
            p = np.sort(sm[:, 1:5], axis=1)
            ok &= bool((np.abs(p[:, -1] - 0.4) > 1e-3).all() and (p[:, -1] - p[:, -2] > 1e-3).all())
            q.add(sm)
            if D.ctc_predict(D.ctc_decode2(np.concatenate(q.get_all(), 0), 6), "12"):
                want.append(ci)
                q.clear()
                state[:] = 0
            pos += n
        if ok:
            assert [ci for ci, f in enumerate(fired) if s in f] == want


def test_frontend_argument_errors():
    from keyword_spotting_amd import _lib, get_config
    from keyword_spotting_amd.frontend import MelFrontend
    with pytest.raises(_lib.UnsupportedError):
        MelFrontend(get_config(fft_size=410))
    with pytest.raises(_lib.InvalidArgumentError):
        MelFrontend(get_config(fmax=9000))
    cfg, fe = _frontend()
    with pytest.raises(_lib.InvalidArgumentError, match="rank 2"):
        fe.forward(torch.zeros(2, 3, 400))


def test_stream_manager_accepts_int16_pcm_like_the_ring_buffer():
    """detector.py:40-43,74-79: the sound card's int16 samples are scaled by 2^-15 -- here on the device; the
    result equals feeding the host-converted floats, chunk by chunk, bit for bit."""
    from keyword_spotting_amd.detector import StreamManager, buf_to_float
    from keyword_spotting_amd.rnn_ctc import DeployModel
    cfg, fe = _frontend()
    w = G.init_weights()
    w["Wfc"] = (w["Wfc"] * 3).astype(np.float32)
    rng = np.random.default_rng(221)
    pcm16 = (rng.standard_normal((5, 3600 * 6)) * 4000).clip(-32768, 32767).astype(np.int16)
    as_float = (1.0 / 32768.0) * pcm16.astype(np.float32)                       # buf_to_float of the reference
    np.testing.assert_array_equal(buf_to_float(torch.from_numpy(pcm16)).numpy(), as_float)
    a, b = StreamManager(DeployModel(cfg, w), 5, label="12"), StreamManager(DeployModel(cfg, w), 5, label="12")
    for c in range(6):
        ha = a.feed_pcm(torch.from_numpy(pcm16[:, 3600 * c:3600 * (c + 1)]).cuda(), fe).clone()
        hb = b.feed_pcm(torch.from_numpy(as_float[:, 3600 * c:3600 * (c + 1)]), fe).clone()
        assert torch.equal(ha, hb) and torch.equal(a.state, b.state)
    with pytest.raises(ValueError):
        buf_to_float(torch.zeros(4, dtype=torch.int32))


def test_carry_form_equals_concatenation_and_streams_like_the_host_loop():
    """kws_frontend_run_carry reads [carry | chunk] in place: same mel bits as the materialised concatenation, same
    next carry as data[-res:] (detector.py:179-183); StreamManager.feed_pcm built on it fires like HotwordDetector."""
    from keyword_spotting_amd.detector import HotwordDetector, StreamManager
    from keyword_spotting_amd.rnn_ctc import DeployModel
    cfg, fe = _frontend()
    rng = np.random.default_rng(231)
    for b, nc, n, keep in ((3, 240, 3600, 320), (5, 320, 3600, 240), (2, 0, 3600, 240), (4, 399, 1, 400), (1, 100, 50, 150)):
        carry = torch.from_numpy(rng.standard_normal((b, nc)).astype(np.float32)).cuda()
        chunk = torch.from_numpy(rng.standard_normal((b, n)).astype(np.float32)).cuda()
        mel, nxt = fe.forward_carry(carry if nc else None, chunk, keep)
        data = torch.cat([carry, chunk], 1)
        assert torch.equal(nxt, data[:, data.shape[1] - keep:])
        if nc + n >= cfg.fft_size:
            assert torch.equal(mel, fe.forward(data))
        else:
            assert mel.shape[1] == 0
    w = G.init_weights()
    w["Wfc"] = (w["Wfc"] * 3).astype(np.float32)
    det = HotwordDetector(DeployModel(cfg, w), batch=3, label="12")
    mgr = StreamManager(DeployModel(cfg, w), 3, label="12")
    pcm = (rng.standard_normal((3, 40000)) * 0.2).astype(np.float32)
    pos = 0
    for n in (200, 150, 3600, 3600, 1000, 5000, 3600, 3600, 3600, 3600, 3600, 3600):   # incl. chunks shorter than a frame
        piece = pcm[:, pos:pos + n]
        pos += n
        want = np.zeros(3, np.int32)
        want[det.feed_pcm(piece, fe)] = 1
        got = mgr.feed_pcm(torch.from_numpy(piece), fe).cpu().numpy()
        np.testing.assert_array_equal(got, want)
        assert torch.equal(mgr.state, det.state)
    # long chunks: window rows wider than a wave (max_frames 120 -> 128-byte rows), 5 chunks queued: the window walk takes ten
    # 64-cell trips with rows straddling trips, and every chunk length differs
    det = HotwordDetector(DeployModel(cfg, w), batch=3, label="1", window_chunks=5)
    mgr = StreamManager(DeployModel(cfg, w), 3, label="1", window_chunks=5, max_frames=120)
    pcm = (rng.standard_normal((3, 150000)) * 0.2).astype(np.float32)
    pos, fired = 0, 0
    for n in (16000, 19000, 5000, 17777, 3600, 18000, 16001, 12000, 9000, 15000):
        piece = pcm[:, pos:pos + n]
        pos += n
        want = np.zeros(3, np.int32)
        want[det.feed_pcm(piece, fe)] = 1
        got = mgr.feed_pcm(torch.from_numpy(piece), fe).cpu().numpy()
        np.testing.assert_array_equal(got, want)
        fired += int(got.sum())
    assert fired > 0


def test_session_run_takes_the_pcm_feed_of_the_shipped_graph():
    """models/rnn_ctc.py:130-165 / detector.py:190-193: `model/inputX:0` is the 1-D PCM chunk; the graph frames it,
    takes |rfft|, projects on the mel basis and runs the GRU stack.  Checked against frontend_oracle + gru_oracle,
    chunk after chunk with the sample carry of detector.py:179-183 and the state threaded through."""
    from keyword_spotting_amd import get_config
    from keyword_spotting_amd.rnn_ctc import DeployModel
    from keyword_spotting_amd import _lib
    cfg = get_config()
    w = G.init_weights()
    m = DeployModel(cfg, w)
    rng = np.random.default_rng(241)
    pcm = (rng.standard_normal(3600 * 4) * 0.3).astype(np.float32)
    state = np.zeros((2, 1, 128), np.float32)
    res = pcm[:0]
    ostate = np.zeros((2, 1, 128), np.float64)
    for c in range(4):
        data = np.concatenate((res, pcm[3600 * c:3600 * (c + 1)]), 0)            # detector.py:179
        keep = (len(data) - 400) % 160 + 240                                      # :181-182
        res = data[-keep:]                                                        # :183
        softmax, logit, state = m.run(["model/softmax:0", "model/logit:0", "model/rnn_states:0"],
                                      {"model/inputX:0": data, "model/rnn_initial_states:0": state})
        t = D.frames_in(len(data))
        assert tuple(softmax.shape) == (t, 6) and tuple(logit.shape) == (1, t, 6) and tuple(state.shape) == (2, 1, 128)
        mel = F.melspec(data[None], n_mels=40)
        want_l, ostate = G.gru_forward(w, mel, ostate, dtype=np.float64)
        scale = max(1.0, float(np.abs(want_l).max()))
        assert np.abs(logit.cpu().numpy() - want_l).max() < 1e-4 * scale
        assert np.abs(softmax.cpu().numpy() - G.softmax(want_l)[0]).max() < 2e-5
        assert np.abs(state.cpu().numpy() - ostate).max() < 1e-4
    # fewer samples than one frame: zero frames, state untouched (tf_frame, utils/stft.py:27-81)
    sm, st = m.run(["model/softmax:0", "model/rnn_states:0"],
                   {"model/inputX:0": pcm[:399], "model/rnn_initial_states:0": state})
    assert tuple(sm.shape) == (0, 6) and torch.equal(st, state)
    with pytest.raises(_lib.InvalidArgumentError):                               # float32 placeholder
        m.run("model/softmax:0", {"model/inputX:0": np.zeros(3600, np.int16), "model/rnn_initial_states:0": state})
    with pytest.raises(_lib.InvalidArgumentError):
        m.run("model/softmax:0", {"model/inputX:0": np.zeros((1, 2, 3, 40), np.float32), "model/rnn_initial_states:0": state})


def test_int16_pcm_gives_the_same_decisions_in_both_detector_classes():
    """RingBuffer.get (detector.py:74-79) hands the loop int16 PCM scaled by 2^-15: HotwordDetector.feed_pcm and
    StreamManager.feed_pcm convert it the same way, so VAD decisions, mel and triggers agree."""
    from keyword_spotting_amd.detector import HotwordDetector, StreamManager
    from keyword_spotting_amd.rnn_ctc import DeployModel
    cfg, fe = _frontend()
    w = G.init_weights()
    w["Wfc"] = (w["Wfc"] * 3).astype(np.float32)
    rng = np.random.default_rng(251)
    pcm16 = (rng.standard_normal((4, 3600 * 8)) * 3000).clip(-32768, 32767).astype(np.int16)
    pcm16[1, 3600 * 2:3600 * 4] = (pcm16[1, 3600 * 2:3600 * 4] // 4000)           # near-silent chunks: sum|x| < 30
    det = HotwordDetector(DeployModel(cfg, w), batch=4, label="12")
    mgr = StreamManager(DeployModel(cfg, w), 4, label="12")
    pos = 0
    for n in (100, 250, 3600, 3601, 3599, 1, 3600, 3600, 3600, 3600):       # incl. less than one frame in total, odd lengths
        piece = pcm16[:, pos:pos + n]
        pos += n
        want = np.zeros(4, np.int32)
        want[det.feed_pcm(torch.from_numpy(piece), fe)] = 1
        got = mgr.feed_pcm(torch.from_numpy(piece), fe).cpu().numpy()
        np.testing.assert_array_equal(got, want)
        assert torch.equal(mgr.state, det.state)


@pytest.mark.parametrize("seed,as_int16", [(261, False), (262, True)])
def test_native_loop_random_chunk_lengths_equal_the_host_mirror(seed, as_int16):
    """kws_stream_feed against the HotwordDetector mirror of detector.py:158-209 over 50 chunks of random length
    (1 .. 5000 samples: sub-frame chunks, odd lengths, chunks that yield 0 .. 31 frames), a label that fires, silent
    stretches that trip the VAD reset -- hits and recurrent state identical after every chunk."""
    from keyword_spotting_amd.detector import HotwordDetector, StreamManager
    from keyword_spotting_amd.rnn_ctc import DeployModel
    cfg, fe = _frontend()
    w = G.init_weights(seed=3)
    w["Wfc"] = (w["Wfc"] * 3).astype(np.float32)
    rng = np.random.default_rng(seed)
    b = 7
    # a one-word label this random model really emits on noise (random weights never spell "1233")
    probe = DeployModel(cfg, w)
    sm = probe.forward(fe.forward(torch.from_numpy((rng.standard_normal((b, 8000)) * 0.2).astype(np.float32))),
                       probe.zero_state(b), want_logits=False)["softmax"].cpu().numpy()
    words = np.concatenate([D.ctc_decode2(sm[k], 6)[1::2] for k in range(b)])
    assert words.size > 0
    label = str(int(np.bincount(words).argmax()))
    det = HotwordDetector(DeployModel(cfg, w), batch=b, label=label)
    mgr = StreamManager(DeployModel(cfg, w), b, label=label)
    fired = 0
    for c in range(50):
        n = int(rng.choice([rng.integers(1, 400), rng.integers(400, 5001), 3600]))
        x = rng.standard_normal((b, n)) * 0.2
        quiet = rng.random(b) < 0.15
        x[quiet] *= 1e-4                                             # sum|x| far below the VAD threshold of 30
        if as_int16:
            piece = torch.from_numpy((x * 32768).clip(-32768, 32767).astype(np.int16))
        else:
            piece = torch.from_numpy(x.astype(np.float32))
        want = np.zeros(b, np.int32)
        want[det.feed_pcm(piece, fe)] = 1
        got = mgr.feed_pcm(piece, fe).cpu().numpy()
        np.testing.assert_array_equal(got, want, err_msg="chunk %d of %d samples" % (c, n))
        assert torch.equal(mgr.state, det.state), (c, n)
        fired += int(want.sum())
    assert fired > 0


def _oracle_loop(w, chunks, label, window=15, vad_thres=30):
    """detector.py:158-209 replayed with the oracle pieces only (numpy, fp64 model): per chunk -> (fired, the state the
    model returned on that chunk, no frame near a decode threshold)."""
    res = np.zeros(0, np.float32)
    state = np.zeros((2, 1, 128), np.float64)
    q = D.SimpleQueue(window)
    out = []
    for data in chunks:
        if len(data) == 0:                                            # :164-166
            out.append((False, state.copy(), True))
            continue
        if not D.vad(data, vad_thres):                                # :168-177
            state[:] = 0
            q.clear()
        data = np.concatenate((res, data), 0)                         # :179
        keep = (len(data) - 400) % 160 + 240                          # :181-182
        res = data[-keep:]                                            # :183 (all of it when len(data) < 400)
        margin_ok = True
        if D.frames_in(len(data)) > 0:
            lg, state = G.gru_forward(w, F.melspec(data[None], n_mels=40).astype(np.float32), state, dtype=np.float64)
            sm = G.softmax(lg)[0]
            p = np.sort(sm[:, 1:5], axis=1)
            margin_ok = bool((np.abs(p[:, -1] - 0.4) > 1e-3).all() and (p[:, -1] - p[:, -2] > 1e-3).all())
        else:
            sm = np.zeros((0, 6))                                     # sess.run over zero frames
        q.add(sm)                                                     # :195 -- an empty softmax still takes a slot
        fired = bool(D.ctc_predict(D.ctc_decode2(np.concatenate(q.get_all(), 0), 6), label))   # :197-201
        seen = state.copy()              # what sess.run returned: the device applies the trigger's clean_state() lazily,
        if fired:                        # as a reset mask on the next run, so this is what its state buffer holds now
            q.clear()                                                 # :203
            state[:] = 0                                              # :208
        out.append((fired, seen, margin_ok))
    return out


@pytest.mark.parametrize("as_int16", [False, True])
def test_sub_frame_chunks_follow_the_reference_loop_against_the_oracle(as_int16):
    """detector.py:168-177 runs on EVERY non-empty chunk, also one that is shorter than a frame: a silent 100-sample read
    after speech zeroes the state and clears the window before its samples are carried; a speech one changes nothing but
    still pushes an empty softmax into the 15-slot window.  Native loop and host mirror against an oracle-only replay."""
    from keyword_spotting_amd.detector import HotwordDetector, StreamManager
    from keyword_spotting_amd.rnn_ctc import DeployModel
    cfg, fe = _frontend()
    w = G.init_weights(seed=3)
    w["Wfc"] = (w["Wfc"] * 3).astype(np.float32)
    rng = np.random.default_rng(271)
    b = 3
    probe = DeployModel(cfg, w)
    sm = probe.forward(fe.forward(torch.from_numpy((rng.standard_normal((b, 8000)) * 0.2).astype(np.float32))),
                       probe.zero_state(b), want_logits=False)["softmax"].cpu().numpy()
    words = np.concatenate([D.ctc_decode2(sm[k], 6)[1::2] for k in range(b)])
    label = str(int(np.bincount(words).argmax()))
    # (length, which streams are silent): sub-frame silent after speech, sub-frame speech, empty read, silent full chunk
    # (a 3600-sample chunk leaves 240..399 samples carried, so "shorter than a frame in total" means a few dozen samples)
    plan = [(3600, []), (100, [0]), (3600, []), (50, []), (0, []), (3600, [1]), (60, [2]), (1, []), (3600, []), (30, [0, 1, 2]),
            (3600, [])] + [(5, [])] * 17 + [(3600, [])]
    carried, sub_frame = 0, []
    for n, _ in plan:
        total = carried + n
        sub_frame.append(n > 0 and total < 400)
        carried = total if total < 400 else (total - 400) % 160 + 240
    assert sub_frame[1] and sub_frame[3] and sub_frame[7] and sum(sub_frame) >= 15
    chunks = []
    for n, quiet in plan:
        x = rng.standard_normal((b, n)) * 0.2
        x[quiet] *= 1e-4
        if as_int16:
            x = (x * 32768).clip(-32768, 32767).astype(np.int16)
        else:
            x = x.astype(np.float32)
        chunks.append(x)
    as_float = [c.astype(np.float32) / 32768.0 if as_int16 else c for c in chunks]
    want = [_oracle_loop(w, [c[s] for c in as_float], label) for s in range(b)]
    det = HotwordDetector(DeployModel(cfg, w), batch=b, label=label)
    mgr = StreamManager(DeployModel(cfg, w), b, label=label)
    ok = [True] * b
    checked = fired = 0
    for ci, piece in enumerate(chunks):
        hits_d = np.zeros(b, np.int32)
        hits_d[det.feed_pcm(torch.from_numpy(piece), fe)] = 1
        hits_m = mgr.feed_pcm(torch.from_numpy(piece), fe).cpu().numpy()
        np.testing.assert_array_equal(hits_m, hits_d, err_msg="chunk %d" % ci)
        assert torch.equal(mgr.state, det.state), ci
        st = mgr.state.cpu().numpy()
        for s in range(b):
            f, ostate, margin = want[s][ci]
            ok[s] = ok[s] and margin
            if not ok[s]:
                continue                         # a frame within 1e-3 of a decode threshold: fp32 vs fp64 may decide differently
            assert bool(hits_m[s]) == f, (ci, s)
            assert np.abs(st[:, s] - ostate[:, 0]).max() < 1e-4, (ci, s)
            checked += 1
            fired += int(f)
        if ci == 1:
            assert not mgr.state[:, 0].any()      # the silent sub-frame read zeroed stream 0 at once, not one chunk later
    assert checked > 2 * len(plan) and fired > 0


def test_stream_handle_outliving_its_model_or_frontend_fails_cleanly():
    """kws_stream borrows the model, front-end and window handles: feeding after one of them was destroyed is an
    InvalidArgument error, not a use of freed memory."""
    import ctypes
    from keyword_spotting_amd import _lib
    from keyword_spotting_amd.detector import StreamManager
    from keyword_spotting_amd.frontend import MelFrontend
    from keyword_spotting_amd.rnn_ctc import DeployModel
    cfg, fe = _frontend()
    model = DeployModel(cfg, G.init_weights())
    mgr = StreamManager(model, 2)
    pcm = torch.zeros(2, 3600)
    mgr.feed_pcm(pcm, fe)
    stream = mgr._stream
    lib = _lib.load()
    hit = torch.zeros(2, dtype=torch.int32, device="cuda")
    fe.close()
    rc = lib.kws_stream_feed(stream, _lib.ptr(pcm.cuda()), 3600, 0, _lib.ptr(hit), None)
    assert rc == _lib.KWS_ERR_INVALID_ARGUMENT and b"destroyed" in lib.kws_last_error()
    with pytest.raises(_lib.InvalidArgumentError):
        mgr.feed_pcm(pcm, fe)                                 # the Python wrapper notices the closed front-end itself
    fe2 = MelFrontend(cfg)
    mgr.feed_pcm(pcm, fe2)                                    # a new front-end: the stream is recreated
    model.close()
    with pytest.raises(_lib.InvalidArgumentError):
        mgr.feed_pcm(pcm, fe2)


@pytest.mark.parametrize("fft,hop,n_mel", [(256, 128, 40), (320, 160, 24), (480, 160, 60), (64, 16, 8)])
def test_other_frame_lengths_run_the_dense_dft_kernel(fft, hop, n_mel):
    """Only the reference's 400-sample frames take the FFT kernel; every other multiple of 16 keeps the dense-DFT kernel
    (frontend_kernels.hip) -- same oracle, same tolerance."""
    cfg, fe = _frontend(fft_size=fft, hop_size=hop, n_mel=n_mel)
    rng = np.random.default_rng(300 + fft)
    pcm = (rng.standard_normal((3, 5 * fft + 77)) * 0.1).astype(np.float32)
    got = fe.forward(torch.from_numpy(pcm)).cpu().numpy()
    want = F.melspec(pcm, n_fft=fft, hop=hop, n_mels=n_mel)
    assert got.shape == want.shape and want.shape[1] == 1 + (pcm.shape[1] - fft) // hop
    assert np.abs(got - want).max() < 2e-5 * np.abs(want).max()


@pytest.mark.parametrize("n_mel", [40, 60, 13])
def test_fft_kernel_and_dense_kernel_agree_on_400_sample_frames(n_mel, monkeypatch):
    """The 16 x 25 FFT (fft_frontend.hip) and the dense DFT it replaced compute the same mel spectrogram: both within the
    oracle tolerance, and within it of each other -- incl. a filter count that is not a multiple of 4 (scalar stores),
    a batch whose frame count is not a multiple of the 16-frame workgroup tile, and the carry seam."""
    from keyword_spotting_amd import get_config
    from keyword_spotting_amd.frontend import MelFrontend
    cfg = get_config(n_mel=n_mel)
    fast = MelFrontend(cfg)
    monkeypatch.setenv("KWS_FRONTEND_DENSE", "1")
    dense = MelFrontend(cfg)
    monkeypatch.delenv("KWS_FRONTEND_DENSE")
    rng = np.random.default_rng(310 + n_mel)
    for batch, n in ((7, 3600), (1, 400), (3, 400 + 160 * 16), (5, 6000)):
        pcm = (rng.standard_normal((batch, n)) * 0.1).astype(np.float32)
        a = fast.forward(torch.from_numpy(pcm)).cpu().numpy()
        b = dense.forward(torch.from_numpy(pcm)).cpu().numpy()
        want = F.melspec(pcm, n_mels=n_mel)
        scale = np.abs(want).max()
        assert np.abs(a - want).max() < 2e-5 * scale and np.abs(b - want).max() < 2e-5 * scale
        assert np.abs(a - b).max() < 2e-5 * scale
    carry = torch.from_numpy(rng.standard_normal((4, 333)).astype(np.float32)).cuda()
    chunk = torch.from_numpy(rng.standard_normal((4, 3600)).astype(np.float32)).cuda()
    ma, na = fast.forward_carry(carry, chunk, 300)
    mb, nb = dense.forward_carry(carry, chunk, 300)
    assert torch.equal(na, nb) and (ma - mb).abs().max() < 2e-5 * mb.abs().max()
    assert torch.equal(ma, fast.forward(torch.cat([carry, chunk], 1)))          # seam path == contiguous path, bit for bit


def test_frontend_full_size_chunk_properties():
    """4096 streams x one 225 ms chunk with carried samples (the bench shape): sampled streams against the oracle, every
    stream's result independent of the batch it sits in (bitwise), linear in the input scale, finite."""
    cfg, fe = _frontend()
    gen = torch.Generator(device="cuda").manual_seed(5)
    carry = torch.randn(4096, 240, generator=gen, device="cuda") * 0.1
    chunk = torch.randn(4096, 3600, generator=gen, device="cuda") * 0.1
    mel, nxt = fe.forward_carry(carry, chunk, 240)
    assert mel.shape == (4096, 22, 40) and bool(torch.isfinite(mel).all())
    data = torch.cat([carry, chunk], 1)
    assert torch.equal(nxt, data[:, -240:])
    pick = [0, 1, 15, 16, 17, 2047, 2048, 4094, 4095]
    want = F.melspec(data[pick].cpu().numpy(), n_mels=40)
    got = mel[pick].cpu().numpy()
    assert np.abs(got - want).max() < 2e-5 * np.abs(want).max()
    sub = fe.forward(data[pick].contiguous())                   # other batch composition, other tiles, no seam path
    assert torch.equal(sub, mel[pick])
    twice, _ = fe.forward_carry(carry * 2, chunk * 2, 240)      # |rfft| and the projection are homogeneous of degree 1: exact in fp32
    assert torch.equal(twice, mel * 2)
