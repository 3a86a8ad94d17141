"""Soak / determinism: inline-asm MFMA chains are opaque to hipcc's hazard recogniser, and a missing wait state
shows up as RARE, timing-dependent errors (cold caches, first launches).  Repeated launches must be bit-identical,
fresh handles included, and many random shapes must agree across kernel families and with the oracle."""
import numpy as np
import pytest
import torch

from oracle import gru_oracle as G

pytestmark = pytest.mark.gpu


def _model(w, n_mel=40, layers=2, kernel="auto", precision="fp32"):
    from keyword_spotting_amd import get_config
    from keyword_spotting_amd.rnn_ctc import DeployModel
    return DeployModel(get_config(n_mel=n_mel, num_layers=layers, precision=precision), w, kernel=kernel)


@pytest.mark.parametrize("precision,layers", [("fp32", 2), ("fp32", 1), ("bf16", 2), ("bf16", 1), ("f16x3", 2), ("f16x3", 1), ("f16x3", 3)])
def test_fresh_handles_and_repeats_are_bit_identical(precision, layers):
    w = G.random_weights(40, 128, layers, 6, seed=301)
    rng = np.random.default_rng(302)
    ref = {}
    for rep in range(12):
        m = _model(w, layers=layers, precision=precision)          # fresh handle: cold weights every time
        for (b, t) in ((5, 2), (33, 3), (16, 17), (4, 1)):
            mel = G.synthetic_mel(b, t, 40, seed=303 + b + t)
            st0 = (0.5 * np.random.default_rng(304 + b).standard_normal((layers, b, 128))).astype(np.float32)
            r = m.forward(torch.from_numpy(mel), torch.from_numpy(st0))
            key = (b, t)
            got = (r["logits"].cpu().numpy(), r["state"].cpu().numpy())
            if key not in ref:
                ref[key] = got
            else:
                assert np.array_equal(got[0], ref[key][0]) and np.array_equal(got[1], ref[key][1]), (rep, key)
        m.close()


def test_random_shapes_resident_vs_generic_vs_oracle():
    rng = np.random.default_rng(311)
    for trial in range(25):
        layers = int(rng.integers(1, 4))
        n_mel = int(rng.choice([32, 40, 48, 60, 64]))
        b, t = int(rng.integers(1, 70)), int(rng.integers(1, 40))
        w = G.random_weights(n_mel, 128, layers, 6, seed=400 + trial)
        mel = G.synthetic_mel(b, t, n_mel, seed=500 + trial)
        st0 = (0.5 * rng.standard_normal((layers, b, 128))).astype(np.float32)
        lens = rng.integers(0, t + 1, b).astype(np.int32)
        want_l, want_s = G.gru_forward(w, mel, st0, seq_len=lens, dtype=np.float64)
        outs = []
        for kernel in ("resident", "generic"):
            r = _model(w, n_mel, layers, kernel).forward(torch.from_numpy(mel), torch.from_numpy(st0),
                                                         seq_len=torch.from_numpy(lens))
            outs.append(r)
            assert np.abs(r["logits"].cpu().numpy() - want_l).max() < 1e-4, (trial, kernel)
            assert np.abs(r["state"].cpu().numpy() - want_s).max() < 1e-4, (trial, kernel)
        assert (outs[0]["logits"] - outs[1]["logits"]).abs().max().item() < 3e-5


def test_two_handles_on_two_streams_do_not_interfere():
    """All work is enqueued on the caller's stream and scratch is per handle: two models stepping concurrently on two
    HIP streams (different batch / chunk shapes, scratch re-grown mid-way) give the bits of running them alone."""
    import torch
    from keyword_spotting_amd import get_config
    from keyword_spotting_amd.rnn_ctc import DeployModel
    from oracle import gru_oracle as G
    cfg = get_config()
    wa, wb = G.random_weights(40, 128, 2, 6, seed=401), G.random_weights(40, 128, 2, 6, seed=402)
    mela = torch.from_numpy(G.synthetic_mel(200, 64, 40, seed=403)).cuda()
    melb = torch.from_numpy(G.synthetic_mel(77, 90, 40, seed=404)).cuda()

    def run(model, mel, chunks, stream=None):
        st, pos, outs = model.zero_state(mel.shape[0]), 0, []
        for n in chunks:
            if stream is None:
                r = model.forward(mel[:, pos:pos + n].contiguous(), st)
            else:
                with torch.cuda.stream(stream):
                    r = model.forward(mel[:, pos:pos + n].contiguous(), st)
            st, pos = r["state"], pos + n
            outs.append(r["logits"])
        return outs, st

    ca, cb = [5, 22, 23, 14], [30, 1, 59]
    ref_a, sa = run(DeployModel(cfg, wa), mela, ca)
    ref_b, sb = run(DeployModel(cfg, wb), melb, cb)
    torch.cuda.synchronize()
    ma, mb = DeployModel(cfg, wa), DeployModel(cfg, wb)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    s1.wait_stream(torch.cuda.current_stream()); s2.wait_stream(torch.cuda.current_stream())
    sta, stb = ma.zero_state(200), mb.zero_state(77)
    got_a, got_b, pa, pb = [], [], 0, 0
    for i in range(max(len(ca), len(cb))):          # interleave the two streams chunk by chunk
        if i < len(ca):
            with torch.cuda.stream(s1):
                r = ma.forward(mela[:, pa:pa + ca[i]].contiguous(), sta)
                sta, pa = r["state"], pa + ca[i]
                got_a.append(r["logits"])
        if i < len(cb):
            with torch.cuda.stream(s2):
                r = mb.forward(melb[:, pb:pb + cb[i]].contiguous(), stb)
                stb, pb = r["state"], pb + cb[i]
                got_b.append(r["logits"])
    torch.cuda.synchronize()
    for x, y in zip(ref_a + ref_b + [sa, sb], got_a + got_b + [sta, stb]):
        assert torch.equal(x, y)


def test_layer_pipelined_launch_equals_sequential_launches():
    """The streaming kernel launches all layers in one grid when L x groups fits the CUs (producer/consumer per frame)
    and layer by layer otherwise; both must give the same bits.  2080 streams x 2 layers = 260 workgroups does not
    fit 256 CUs -> sequential; its first 300 streams alone (19 groups) -> pipelined."""
    import torch
    from keyword_spotting_amd import get_config
    from keyword_spotting_amd.rnn_ctc import DeployModel
    from oracle import gru_oracle as G
    for layers, hidden, n_mel in ((2, 128, 40), (4, 256, 60), (3, 64, 40)):
        cfg = get_config(num_layers=layers, hidden_size=hidden, n_mel=n_mel)
        w = G.random_weights(n_mel, hidden, layers, 6, seed=411)
        big = 16 * (256 // layers + 2)
        mel = torch.from_numpy(G.synthetic_mel(big, 37, n_mel, seed=412)).cuda()
        st = (0.3 * torch.randn(layers, big, hidden, generator=torch.Generator().manual_seed(5))).cuda()
        m = DeployModel(cfg, w, kernel="generic")
        seq = m.forward(mel, st)                                  # too many groups: one launch per layer
        k = 300
        pip = m.forward(mel[:k].contiguous(), st[:, :k].contiguous())     # fits: layer-pipelined
        assert torch.equal(pip["logits"], seq["logits"][:k])
        assert torch.equal(pip["state"], seq["state"][:, :k])
        # chunked through the pipelined path == one shot
        s2, parts, pos = st[:, :k].contiguous(), [], 0
        for n in (1, 22, 14):
            r = m.forward(mel[:k, pos:pos + n].contiguous(), s2)
            s2, pos = r["state"], pos + n
            parts.append(r["logits"])
        assert torch.equal(torch.cat(parts, 1), pip["logits"]) and torch.equal(s2, pip["state"])


def test_no_stale_data_leaks_between_calls():
    """Paths that hand data between CUs inside one kernel (int8: vector stores -> scalar loads; layer-pipelined launch:
    producer -> consumer workgroups) once raced in a way that repeating identical calls cannot see: a stale read returns
    the previous call's identical bytes.  So: a call on DIFFERENT inputs must equal a fresh model's first call."""
    import torch
    from keyword_spotting_amd import get_config, weights
    from keyword_spotting_amd.rnn_ctc import DeployModel
    cases = [("int8", get_config(precision="int8"), 4096, 24, "auto"),
             ("pipelined 4x256", get_config(n_mel=60, hidden_size=256, num_layers=4), 1024, 24, "auto"),
             ("pipelined 8x64", get_config(hidden_size=64, num_layers=8), 512, 24, "auto"),
             ("bf16", get_config(precision="bf16"), 4096, 24, "auto"),
             ("f16x3", get_config(precision="f16x3"), 4096, 24, "auto"),
             ("f16x3, three layers, odd frame count", get_config(precision="f16x3", num_layers=3), 1000, 23, "auto"),
             ("fp32 resident", get_config(), 4096, 24, "auto"),
             ("fp32 resident, layers overlapped on streams", get_config(), 1024, 96, "auto")]
    g = torch.Generator(device="cuda").manual_seed(77)
    for name, cfg, b, t, kernel in cases:
        w = weights.init_weights(cfg, seed=9)
        x1 = (torch.randn(b, t, cfg.n_mel, device="cuda", generator=g).abs() * 2).contiguous()
        x2 = (torch.randn(b, t, cfg.n_mel, device="cuda", generator=g).abs() * 2).contiguous()
        s1 = (0.3 * torch.randn(cfg.num_layers, b, cfg.hidden_size, device="cuda", generator=g)).contiguous()
        s2 = (0.3 * torch.randn(cfg.num_layers, b, cfg.hidden_size, device="cuda", generator=g)).contiguous()
        m = DeployModel(cfg, w, kernel=kernel)
        first = m.forward(x1, s1)
        again = [m.forward(x2, s2) for _ in range(3)]
        fresh = DeployModel(cfg, w, kernel=kernel).forward(x2, s2)
        back = m.forward(x1, s1)
        for r in again:
            assert torch.equal(r["logits"], fresh["logits"]) and torch.equal(r["state"], fresh["state"]), name
        assert torch.equal(back["logits"], first["logits"]) and torch.equal(back["state"], first["state"]), name


def test_layers_overlapped_on_streams_equal_sequential_launches():
    """Long calls on few streams run their layers on separate HIP streams, time-blocked (step_overlapped); long calls on
    many streams run layer after layer.  Same kernels, so the same bits -- including fused tokens, prev_word carry,
    sequence lengths that end inside a block, the reset mask and a frame count that is not a multiple of the block."""
    import torch
    from keyword_spotting_amd import get_config
    from keyword_spotting_amd.rnn_ctc import DeployModel
    from oracle import gru_oracle as G
    for layers in (2, 3):
        cfg = get_config(num_layers=layers)
        w = G.random_weights(40, 128, layers, 6, seed=421)
        w["Wfc"] = (w["Wfc"] * 3).astype(np.float32)
        big, k, t = 16 * (256 // layers + 1), 200, 173
        gen = torch.Generator().manual_seed(422)
        mel = (torch.randn(big, t, 40, generator=gen).abs() * 2).cuda()
        st = (0.3 * torch.randn(layers, big, 128, generator=gen)).cuda()
        seq = torch.randint(0, t + 1, (big,), generator=gen).to(torch.int32)
        seq[:3] = torch.tensor([0, t, 95])
        rst = (torch.rand(big, generator=gen) < 0.3).to(torch.uint8)
        m = DeployModel(cfg, w)
        pw_a, pw_b = m.fresh_prev_word(big), m.fresh_prev_word(k)
        a = m.forward(mel, st, seq_len=seq, reset_mask=rst, prev_word=pw_a)                       # sequential
        b = m.forward(mel[:k].contiguous(), st[:, :k].contiguous(), seq_len=seq[:k], reset_mask=rst[:k], prev_word=pw_b)
        for key in ("logits", "softmax", "tokens"):
            assert torch.equal(b[key], a[key][:k]), (layers, key)
        assert torch.equal(b["state"], a["state"][:, :k]) and torch.equal(pw_b, pw_a[:k])
        assert int((b["tokens"] > 0).sum()) > 0


@pytest.mark.parametrize("precision", ["fp32", "f16x3", "bf16"])
def test_two_host_threads_two_handles(precision):
    """The unit of host concurrency is the handle (kws_amd.h; the reference's Compute is re-entrant,
    octbit/octbit_mat_mul_op.cc:49): two Python threads, each with its own DeployModel and its own HIP stream, step 200
    times side by side (ctypes releases the GIL around kws_step; a barrier per step makes the calls collide) and get the
    bits of running alone on one thread."""
    import threading
    from keyword_spotting_amd import get_config
    from keyword_spotting_amd.rnn_ctc import DeployModel
    cfg = get_config(precision=precision)
    shapes = ((48, 22, 601), (130, 23, 602))                  # (streams, frames, seed): different batch shapes per thread
    ws = [G.random_weights(40, 128, 2, 6, seed=s) for _, _, s in shapes]
    mels = [torch.from_numpy(G.synthetic_mel(b, 3 * t, 40, seed=s + 10)).cuda() for b, t, s in shapes]
    steps = 200

    def drive(i, stream, gate, out, errors):
        try:
            b, t, _ = shapes[i]
            m = DeployModel(cfg, ws[i])
            st, pw = m.zero_state(b), m.fresh_prev_word(b)
            acc_l = torch.zeros(b, t, 6, device="cuda")
            acc_t = torch.zeros(b, t, dtype=torch.int32, device="cuda")
            with torch.cuda.stream(stream):
                for k in range(steps):
                    if gate is not None:
                        gate.wait()
                    mel = mels[i][:, (k % 3) * t:(k % 3 + 1) * t].contiguous()
                    r = m.forward(mel, st, prev_word=pw, state_out=st)
                    acc_l += r["logits"]
                    acc_t += r["tokens"].to(torch.int32)
                stream.synchronize()
            out[i] = (acc_l.cpu(), acc_t.cpu(), st.cpu(), pw.cpu())
            m.close()
        except Exception as exc:          # surfaced by the main thread
            errors.append((i, repr(exc)))
            if gate is not None:
                gate.abort()

    alone, errors = {}, []
    for i in range(2):
        drive(i, torch.cuda.Stream(), None, alone, errors)
    assert not errors, errors
    both, gate = {}, threading.Barrier(2)
    threads = [threading.Thread(target=drive, args=(i, torch.cuda.Stream(), gate, both, errors)) for i in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(300)
    assert not errors and len(both) == 2, errors
    for i in range(2):
        for x, y in zip(alone[i], both[i]):
            assert torch.equal(x, y), (precision, i)


def test_sharing_one_handle_between_threads_is_detected_not_undefined():
    """One handle, two host threads, no lock (unsupported: kws_amd.h): every call either completes with the right bits or
    returns KWS_ERR_BUSY having launched nothing -- never a silently corrupted result.  With the caller's own lock around
    the calls (the supported way to share) all of them succeed, also when the two threads come in on different streams."""
    import threading
    from keyword_spotting_amd import _lib, get_config
    from keyword_spotting_amd.rnn_ctc import DeployModel
    cfg = get_config()
    w = G.random_weights(40, 128, 2, 6, seed=611)
    b, t = 64, 22
    mel = torch.from_numpy(G.synthetic_mel(b, t, 40, seed=612)).cuda()
    m = DeployModel(cfg, w)
    st0 = m.zero_state(b)
    want = m.forward(mel, st0)
    torch.cuda.synchronize()
    want_l, want_s = want["logits"].clone(), want["state"].clone()

    def hammer(lock, stats, bad):
        stream = torch.cuda.Stream()
        with torch.cuda.stream(stream):
            for _ in range(300):
                try:
                    if lock is not None:
                        with lock:
                            r = m.forward(mel, st0)
                            stream.synchronize()            # the caller's lock covers the call AND its kernels
                    else:
                        r = m.forward(mel, st0)
                        stream.synchronize()
                except _lib.BusyError:
                    stats["busy"] += 1
                    continue
                stats["ok"] += 1
                if not (torch.equal(r["logits"], want_l) and torch.equal(r["state"], want_s)):
                    bad.append(1)

    for lock in (None, threading.Lock()):
        stats, bad = {"ok": 0, "busy": 0}, []
        threads = [threading.Thread(target=hammer, args=(lock, stats, bad)) for _ in range(2)]
        for th in threads:
            th.start()
        for th in threads:
            th.join(300)
        assert stats["ok"] + stats["busy"] == 600 and stats["ok"] >= 300
        if lock is not None:
            assert stats["busy"] == 0 and not bad
        else:
            # unlocked: a call that was let in ran alone on the host, but its kernels may still overlap the other thread's
            # next call on the other stream -- kws_step waits for the device when the stream changes, so results stay right
            assert not bad, "%d corrupted results out of %d" % (len(bad), stats["ok"])
    m.close()


def test_a_stream_switch_is_ordered_on_the_device_and_does_not_drain_other_handles():
    """A handle whose calls alternate between two HIP streams is ordered by an event on its own last launch (kws_amd.h) -- not by a
    device-wide wait, which would block the calling thread until every OTHER handle's queue has drained.  Thread B keeps ~100 ms of
    4096 x 300 steps queued on its handle; meanwhile thread A's 24 alternating small calls must return in a fraction of that
    time (loose bound), and both get the bits of running alone."""
    import threading
    import time
    from keyword_spotting_amd import get_config
    from keyword_spotting_amd.rnn_ctc import DeployModel
    cfg = get_config()
    wa, wb = G.random_weights(40, 128, 2, 6, seed=621), G.random_weights(40, 128, 2, 6, seed=622)
    ba, ta, bb, tb, nb = 96, 22, 4096, 300, 32
    mel_a = [torch.from_numpy(G.synthetic_mel(ba, ta, 40, seed=623 + k)).cuda() for k in range(3)]
    mel_b = (torch.randn(bb, tb, 40, device="cuda").abs() * 2).contiguous()

    def a_calls(m, streams):
        st, pw = m.zero_state(ba), m.fresh_prev_word(ba)
        acc = torch.zeros(ba, ta, 6, device="cuda")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(24):
            with torch.cuda.stream(streams[k % len(streams)]):
                r = m.forward(mel_a[k % 3], st, prev_word=pw, state_out=st)
        host_s = time.perf_counter() - t0                   # the calls only: nothing below is timed
        torch.cuda.synchronize()
        return host_s, r["logits"].cpu(), st.cpu(), pw.cpu()

    def b_calls(m, stream, started):
        st = m.zero_state(bb)
        out = {"logits": torch.empty(bb, tb, 6, device="cuda"), "softmax": torch.empty(bb, tb, 6, device="cuda")}
        torch.cuda.synchronize()
        with torch.cuda.stream(stream):
            for k in range(nb):
                m.forward(mel_b, st, state_out=st, out=out)
                if k == 3:
                    started.set()
        return st, out

    # alone
    ma, mb = DeployModel(cfg, wa), DeployModel(cfg, wb)
    mb.reserve(bb, tb)
    _, want_l, want_s, want_p = a_calls(ma, [torch.cuda.current_stream()])
    stb, _ = b_calls(mb, torch.cuda.current_stream(), threading.Event())
    torch.cuda.synchronize()
    want_b = stb.cpu()
    ma.close(); mb.close()
    # together: B's queue is deep while A alternates streams
    ma, mb = DeployModel(cfg, wa), DeployModel(cfg, wb)
    mb.reserve(bb, tb)
    ma.forward(mel_a[0], ma.zero_state(ba))              # A's scratch is sized before the race
    st, pw = ma.zero_state(ba), ma.fresh_prev_word(ba)       # (filled on the default stream: synchronised before the side streams use them)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    started, box = threading.Event(), {}
    sb = torch.cuda.Stream()
    th = threading.Thread(target=lambda: box.setdefault("b", b_calls(mb, sb, started)))
    t0 = time.perf_counter()
    th.start()
    assert started.wait(60)
    ta0 = time.perf_counter()
    for k in range(24):
        with torch.cuda.stream((s1, s2)[k % 2]):
            r = ma.forward(mel_a[k % 3], st, prev_word=pw, state_out=st)
    host_s = time.perf_counter() - ta0
    th.join(120)
    sb.synchronize()
    drain_s = time.perf_counter() - t0                    # B's 32 steps, about 100 ms of device time
    torch.cuda.synchronize()
    assert torch.equal(r["logits"].cpu(), want_l) and torch.equal(st.cpu(), want_s) and torch.equal(pw.cpu(), want_p)
    assert torch.equal(box["b"][0].cpu(), want_b)
    assert drain_s > 0.05, drain_s
    assert host_s < 0.5 * drain_s, "A's calls took %.1f ms of host time while B's queue drained in %.1f ms: a stream switch waited for the device" % (host_s * 1e3, drain_s * 1e3)
    ma.close(); mb.close()


@pytest.mark.parametrize("precision", ["fp32", "f16x3", "bf16"])
def test_a_step_captured_into_a_hip_graph_replays_bit_identically(precision):
    """kws_step launches asynchronously, allocates nothing after kws_reserve and orders stream switches with events, so a 22-frame
    hop can be captured into a hipGraph (torch.cuda.graph) and replayed: 8 replays == 8 eager calls, state carried through the
    graph's own buffers.  The capture stream is NOT the stream of the handle's previous call: the event wait that orders the
    switch is itself captured (include/kws_amd.h: "legal under stream capture")."""
    from keyword_spotting_amd import get_config, weights
    from keyword_spotting_amd.rnn_ctc import DeployModel
    cfg = get_config(precision=precision)
    m = DeployModel(cfg, weights.init_weights(cfg, seed=0))
    b, t = 512, 22
    mel = torch.from_numpy(G.synthetic_mel(b, t, 40, seed=701)).cuda()
    st, pw = m.zero_state(b), m.fresh_prev_word(b)
    out = {"logits": torch.empty(b, t, 6, device="cuda"), "softmax": torch.empty(b, t, 6, device="cuda"),
           "tokens": torch.empty(b, t, dtype=torch.int8, device="cuda")}
    m.reserve(b, t)
    for _ in range(8):
        m.forward(mel, st, prev_word=pw, state_out=st, out=out)            # eager, on the default stream
    torch.cuda.synchronize()
    want = (st.clone(), pw.clone(), out["logits"].clone(), out["tokens"].clone())
    st.zero_(); pw.fill_(-1)
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        m.forward(mel, st, prev_word=pw, state_out=st, out=out)            # first call of the handle on this stream: captured wait on the eager calls' event
    for _ in range(8):
        g.replay()
    torch.cuda.synchronize()
    for x, y in zip((st, pw, out["logits"], out["tokens"]), want):
        assert torch.equal(x, y), precision
    m.close()
