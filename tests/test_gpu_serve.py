"""Several stream managers in flight (the serving shape behind BASELINE.json's "real-time audio streams sustained"): M
StreamManagers of different sizes, fed in turn on two HIP streams, must give every manager the bits it gets when it runs
alone -- trigger decisions chunk by chunk, carried GRU state, restart masks.  What could break it is what the managers
share: the model handle's inter-layer seams and its staging block (one chunk's mel / softmax / masks: kws_model::stage),
ordered across HIP streams by the handle's end-of-call event and nothing else (no device-wide wait: kws_amd.h)."""
import numpy as np
import pytest
import torch

from oracle import decode_oracle as D
from oracle import gru_oracle as G

pytestmark = pytest.mark.gpu


def _emitting(cfg, fe, rng):
    """Random weights whose model says something on noise, and its most frequent word as a one-digit label."""
    from keyword_spotting_amd.rnn_ctc import DeployModel
    b = 48
    noise = torch.from_numpy((rng.standard_normal((b, 16000)) * 0.2).astype(np.float32))
    for seed in range(7200, 7260):
        w = G.random_weights(40, 128, 2, 6, seed=seed)
        w["Wfc"] = (w["Wfc"] * 4.0).astype(np.float32)
        probe = DeployModel(cfg, w)
        sm = probe.forward(fe.forward(noise), probe.zero_state(b), want_logits=False)["softmax"].cpu().numpy()
        probe.close()
        words = np.concatenate([D.ctc_decode2(sm[k], 6)[1::2] for k in range(b)])
        if words.size >= 2 * b:
            return w, str(int(np.bincount(words).argmax()))
    raise AssertionError("no seed gives a model that emits words")


def _chunks(rng, sizes, periods):
    """Per period and manager one int16 PCM chunk of its own (lengths differ from period to period and between managers;
    a few streams silent: VAD reset + window clear)."""
    out = []
    for p in range(periods):
        row = []
        for b in sizes:
            n = int(rng.choice([3600, 3600, 3600, 1800, 5000])) if p else int(rng.choice([200, 3600]))
            pcm = rng.integers(-6000, 6000, (b, n)).astype(np.int16)
            pcm[rng.random(b) < 0.05] //= 4096                    # |x| <= 1: below vad(data, 30)
            row.append(torch.from_numpy(pcm).cuda())
        out.append(row)
    return out


@pytest.mark.parametrize("precision", ["fp32", "f16x3", "bf16"])
def test_managers_interleaved_on_two_streams_equal_each_one_alone(precision):
    from keyword_spotting_amd import get_config
    from keyword_spotting_amd.detector import StreamManager
    from keyword_spotting_amd.frontend import MelFrontend
    from keyword_spotting_amd.rnn_ctc import DeployModel
    cfg = get_config(precision=precision)
    fe = MelFrontend(cfg)
    rng = np.random.default_rng(8100)
    w, label = _emitting(cfg, fe, rng)
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    sizes = (37, 16 * cus + 21, 130)          # the middle one is more groups than CUs: persistent workgroups, window_inc_kernel behind the stack
    periods = 14
    chunks = _chunks(rng, sizes, periods)
    torch.cuda.synchronize()

    def run(assign):
        """assign(i, period) -> (model index, stream index): who serves manager i's chunk of that period."""
        n_models = 1 + max(assign(i, p)[0] for i in range(len(sizes)) for p in range(periods))
        models = [DeployModel(cfg, w) for _ in range(n_models)]
        streams = [torch.cuda.current_stream(), torch.cuda.Stream(), torch.cuda.Stream()]
        mgrs = [StreamManager(models[assign(i, 0)[0]], batch=b, label=label, window_chunks=4) for i, b in enumerate(sizes)]
        torch.cuda.synchronize()
        hits = [[] for _ in sizes]
        for p in range(periods):
            for i, mgr in enumerate(mgrs):
                with torch.cuda.stream(streams[assign(i, p)[1]]):
                    hits[i].append(mgr.feed_pcm(chunks[p][i], fe).clone())        # no host wait between feeds: the calls collide on the device
        torch.cuda.synchronize()
        res = [(torch.stack(hits[i]).cpu(), m.state.cpu(), m.restart.cpu()) for i, m in enumerate(mgrs)]
        for m in mgrs:
            m.close()
        for m in models:
            m.close()
        return res

    ref = run(lambda i, p: (i, 0))                                    # a model handle per manager, one stream: nothing shared
    assert sum(int(r[0].sum()) for r in ref) > 0, "the scenario never triggers"
    cases = {
        "one model, every feed on the other stream than the feed before": lambda i, p: (0, 1 + (i + p) % 2),
        "one model, a stream per manager parity": lambda i, p: (0, 1 + i % 2),
        "two models on two streams": lambda i, p: (i % 2, 1 + i % 2),
    }
    for name, assign in cases.items():
        got = run(assign)
        for i in range(len(sizes)):
            for x, y, what in zip(got[i], ref[i], ("hits", "state", "restart")):
                assert torch.equal(x, y), (precision, name, "manager %d" % i, what)


def test_managers_share_the_models_staging_block_and_own_only_their_state():
    """M managers on one model handle cost M x (state + carry + window summaries): one chunk's intermediates are carved out of
    the handle's staging block, which is as large as the largest manager needs and does not grow with their number."""
    from keyword_spotting_amd import get_config
    from keyword_spotting_amd.detector import StreamManager
    from keyword_spotting_amd.frontend import MelFrontend
    from keyword_spotting_amd.rnn_ctc import DeployModel
    cfg = get_config()
    fe = MelFrontend(cfg)
    m = DeployModel(cfg, G.random_weights(40, 128, 2, 6, seed=8200))
    b = 4096
    pcm = torch.zeros(b, 3600, dtype=torch.int16, device="cuda")
    first = StreamManager(m, batch=b)
    first.feed_pcm(pcm, fe)
    torch.cuda.synchronize()
    allocs0 = m.scratch_stats()[1]
    free0 = torch.cuda.mem_get_info()[0]
    more = [StreamManager(m, batch=b) for _ in range(8)]
    for mgr in more:
        mgr.feed_pcm(pcm, fe)
    torch.cuda.synchronize()
    per_stream = (free0 - torch.cuda.mem_get_info()[0]) / float(8 * b)
    # state 1024 + two carries 2 x 399 x 4 + summaries 15 x 36 + head/count 8 + restart 1 + hit 4 = 4769 B, plus what the
    # allocators round seven blocks per manager up to (2 MiB granules: up to ~0.9 KB per stream at this size); a manager that
    # kept its own chunk intermediates would add 34 frames x (160 + 24) B + masks = 6.3 KB per stream
    # (lower bound: what the library itself allocates -- carries 3192 + summaries 548; the manager's torch tensors may come out
    # of blocks the caching allocator already holds)
    assert 3700 <= per_stream <= 6200, per_stream
    assert m.scratch_stats()[1] == allocs0, "a manager of the same size must not regrow the handle's blocks"
    for mgr in more + [first]:
        mgr.close()
    m.close()


def test_stream_server_periods_equal_plain_managers_and_pacing_reports():
    """keyword_spotting_amd.serving: M managers round-robin on the server's handles / HIP streams give, period by period, the hits
    of M plain StreamManagers fed one after the other on the default stream; run_paced feeds in real time and reports the
    period compute times and deadline misses."""
    from keyword_spotting_amd import get_config
    from keyword_spotting_amd.detector import StreamManager
    from keyword_spotting_amd.frontend import MelFrontend
    from keyword_spotting_amd.rnn_ctc import DeployModel
    from keyword_spotting_amd.serving import StreamServer, run_paced
    cfg = get_config(precision="f16x3")
    fe = MelFrontend(cfg)
    rng = np.random.default_rng(8300)
    w, label = _emitting(cfg, fe, rng)
    s, m_count, periods = 48, 5, 7
    pcm = [[torch.from_numpy(rng.integers(-6000, 6000, (s, 3600)).astype(np.int16)).cuda() for _ in range(m_count)] for _ in range(periods)]
    torch.cuda.synchronize()
    server = StreamServer(cfg, weights=w, streams_per_manager=s, handles=2, label=label, window_chunks=4).resize(m_count)
    assert server.n_streams == s * m_count and len(server.models) == 2
    got = []
    for p in range(periods):
        assert server.feed_period(lambda k: pcm[p][k]) == m_count
        got.append(server.hits().cpu())
    assert server.launches_per_chunk() == 3                     # gate + front-end, two GRU layers with the window step inside the second
    models = [DeployModel(cfg, w) for _ in range(m_count)]
    plain = [StreamManager(models[k], batch=s, label=label, window_chunks=4) for k in range(m_count)]
    total = 0
    for p in range(periods):
        want = torch.stack([plain[k].feed_pcm(pcm[p][k], fe).clone() for k in range(m_count)]).cpu()
        assert torch.equal(got[p], want), p
        total += int(want.sum())
    assert total > 0
    for k in range(m_count):
        assert torch.equal(server.managers[k].state, plain[k].state)
    # shrinking and growing keeps the survivors' state; pacing: 5 periods of 20 ms
    server.resize(3)
    assert server.n_streams == 3 * s and torch.equal(server.managers[2].state, plain[2].state)
    server.resize(4)
    rep = run_paced(server, lambda p, k: pcm[p % periods][k], periods=5, period_s=0.02)
    assert rep["periods"] == 5 and rep["deadline_misses"] == 0 and 0 < rep["compute_ms_p50"] <= rep["compute_ms_max"] < 20.0
    for m in plain:
        m.close()
    for m in models:
        m.close()
    server.close()
