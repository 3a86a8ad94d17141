#!/usr/bin/env python3
"""Repeat-run determinism stress for the paths that synchronise through memory inside a kernel: the layer-pipelined
launch (frame counters in fine-grained memory) and the int8 layer kernel (scalar-cache exchange).  Any coherence or
ordering bug shows up as a run that differs from the first."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from keyword_spotting_amd import get_config, weights
from keyword_spotting_amd.rnn_ctc import DeployModel

def repeat(name, cfg, B, T, n, kernel="auto"):
    m = DeployModel(cfg, weights.init_weights(cfg, seed=3), kernel=kernel)
    mel = (torch.randn(B, T, cfg.n_mel, device="cuda").abs() * 2).contiguous()
    st = (0.3 * torch.randn(cfg.num_layers, B, cfg.hidden_size, device="cuda")).contiguous()
    ref = m.forward(mel, st)
    bad, kinds, first_bad = 0, {}, None
    for i in range(n):
        r = m.forward(mel, st)
        same = torch.equal(r["logits"], ref["logits"]) and torch.equal(r["state"], ref["state"])
        if not same:
            bad += 1
            if first_bad is None:
                d = (r["logits"] - ref["logits"]).abs()
                bs = (d > 0).any(2).any(1).nonzero().flatten()
                first_bad = (i, int((d > 0).sum()), float(d.max()), int((r["state"] != ref["state"]).sum()),
                             "streams %d: %s..%s" % (len(bs), bs[:6].tolist(), bs[-3:].tolist()),
                             "frames", (d > 0).any(2).any(0).nonzero().flatten()[:8].tolist())
        key = (float(r["logits"].double().sum()), float(r["state"].double().sum()))
        kinds[key] = kinds.get(key, 0) + 1
    print("%-44s B=%d T=%d: %d / %d runs differ from the first; %d distinct results %s; first bad %s"
          % (name, B, T, bad, n, len(kinds), sorted(kinds.values(), reverse=True)[:5], first_bad))
    return bad

def repeat_stream(n):
    """kws_stream_feed: the same chunk sequence replayed on fresh managers must give the same hits and state."""
    from keyword_spotting_amd.detector import StreamManager
    from keyword_spotting_amd.frontend import MelFrontend
    import numpy as np
    cfg = get_config()
    w = weights.init_weights(cfg, seed=5)
    w["Wfc"] = (w["Wfc"] * 3).astype(np.float32)
    fe = MelFrontend(cfg)
    B = 4096
    pcm = (torch.randn(B, 3600 * 6, device="cuda") * 0.2 * 32768).clamp(-32768, 32767).to(torch.int16)
    ref, bad = None, 0
    for i in range(max(2, n // 10)):
        mgr = StreamManager(DeployModel(cfg, w), B, label="12")
        hits = [mgr.feed_pcm(pcm[:, 3600 * c:3600 * (c + 1)], fe).clone() for c in range(6)]
        got = (torch.stack(hits), mgr.state.clone())
        if ref is None:
            ref = got
        elif not (torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1])):
            bad += 1
        mgr.close()
    print("%-44s B=%d: %d / %d replays differ; %d hits" % ("kws_stream_feed, int16 PCM, 6 chunks", B, bad, max(2, n // 10) - 1, int(ref[0].sum())))
    return bad

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
tot = 0
tot += repeat_stream(n)
tot += repeat("pipelined 4xGRU h=256 (configs[4])", get_config(n_mel=60, hidden_size=256, num_layers=4), 1024, 60, n)
tot += repeat("pipelined f16x3 4xGRU h=256 (configs[4])", get_config(n_mel=60, hidden_size=256, num_layers=4, precision="f16x3"), 1024, 60, n)
tot += repeat("pipelined f16x3 4xGRU h=256, 3 groups", get_config(n_mel=60, hidden_size=256, num_layers=4, precision="f16x3"), 40, 33, n)
tot += repeat("f16x3 h=256 per-layer launches, ragged", get_config(n_mel=60, hidden_size=256, num_layers=4, precision="f16x3"), 1100, 23, n)
tot += repeat("pipelined 2xGRU h=128 generic", get_config(), 2048, 60, n, kernel="generic")
tot += repeat("pipelined 8xGRU h=64", get_config(hidden_size=64, num_layers=8), 512, 60, n)
tot += repeat("int8 graph", get_config(precision="int8"), 4096, 40, n)
tot += repeat("int8 graph, ragged batch", get_config(precision="int8"), 1000, 23, n)
tot += repeat("bf16 stack", get_config(precision="bf16"), 4096, 100, n)
tot += repeat("f16x3 stack", get_config(precision="f16x3"), 4096, 100, n)
tot += repeat("f16x3 stack, persistent over 513 groups, ragged", get_config(precision="f16x3"), 8200, 23, n)
tot += repeat("fp32 resident", get_config(), 4096, 100, n)
tot += repeat("fp32 resident, layers overlapped on streams", get_config(), 1024, 200, n)
tot += repeat("fp32 resident 4 layers, overlapped", get_config(num_layers=4), 1024, 130, n)
tot += repeat("fp32 resident, persistent over 513 groups", get_config(), 8200, 30, n)
tot += repeat("bf16 stack, persistent over 513 groups", get_config(precision="bf16"), 8200, 30, n)
sys.exit(1 if tot else 0)
