"""Host-side logic that needs no GPU: weight containers, config, window queue, sharding."""
import os
import numpy as np
import pytest

from oracle import gru_oracle as G


def test_config_defaults_and_overrides():
    from keyword_spotting_amd import get_config
    c = get_config()
    assert (c.n_mel, c.hidden_size, c.num_layers, c.num_classes) == (40, 128, 2, 6)
    assert (c.fft_size, c.hop_size, c.samplerate, c.label_seqs) == (400, 160, 16000, "1233")
    c = get_config(n_mel=60, hidden_size="256", num_layers=4)
    assert (c.freq_size, c.hidden_size, c.num_layers) == (60, 256, 4)
    with pytest.raises(AttributeError):
        get_config(nope=1)


def test_blob_roundtrip_and_same_init_as_oracle():
    from keyword_spotting_amd import get_config, weights
    for kw in (dict(), dict(n_mel=60, hidden_size=256, num_layers=4)):
        cfg = get_config(**kw)
        w = weights.init_weights(cfg, seed=3)
        blob = weights.to_blob(cfg, w)
        w2 = weights.from_blob(cfg, blob)
        np.testing.assert_array_equal(weights.to_blob(cfg, w2), blob)
        wo = G.init_weights(cfg.n_mel, cfg.hidden_size, cfg.num_layers, cfg.num_classes, seed=3)
        np.testing.assert_array_equal(G.weights_to_blob(wo), blob)
    assert (w["layers"][0]["bg"] == 1).all() and (w["layers"][0]["bc"] == 0).all()   # TF GRUCell init
    assert np.abs(w["Wfc"]).max() <= 2.0
    with pytest.raises(ValueError):
        weights.from_blob(cfg, blob[:-1])
    bad = weights.init_weights(cfg)
    bad["layers"][1]["Wc"] = bad["layers"][1]["Wc"][:, :-1]
    with pytest.raises(ValueError):
        weights.to_blob(cfg, bad)


def test_npz_roundtrip(tmp_path):
    from keyword_spotting_amd import get_config, weights
    cfg = get_config()
    w = weights.init_weights(cfg, seed=4)
    path = str(tmp_path / "w.npz")
    weights.save_npz(path, w)
    np.testing.assert_array_equal(weights.to_blob(cfg, weights.load_npz(path)), weights.to_blob(cfg, w))


def test_product_queue_matches_reference_trace(golden):
    from keyword_spotting_amd.queue import SimpleQueue
    q = SimpleQueue(15)
    for k, op in enumerate(golden["q_ops"]):
        q.add(k) if op == 0 else q.clear()
        content = q.get_all()
        assert (q.len, int(q.full()), len(content), content[0] if content else -1) == \
            (golden["q_len"][k], golden["q_full"][k], golden["q_n"][k], golden["q_head"][k])


def test_shard_bounds_partition():
    from keyword_spotting_amd.sharding import shard_bounds, shard_seed
    for total in (0, 1, 7, 4096, 32768, 32771):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert shard_bounds(32768, 3, 8) == (12288, 16384)
    assert shard_seed(1, 5) == 6
    with pytest.raises(ValueError):
        shard_bounds(10, 2, 2)


def test_chunk_framer_matches_reference_arithmetic():
    """detector.py:179-183 sample carry: pinned by hand-derived counts and by the oracle restatement."""
    from keyword_spotting_amd.detector import ChunkFramer
    from oracle import decode_oracle as D
    fr = ChunkFramer()
    assert [fr.push(3600) for _ in range(9)] == [21, 22, 23, 22, 23, 22, 23, 22, 23]
    for sizes in ([3600] * 20, [1234, 4000, 800, 160, 159, 4001], [400], [399, 1]):
        fr = ChunkFramer()
        assert [fr.push(n) for n in sizes] == D.chunk_frame_counts(sizes)


def test_tf_variable_names_round_trip(tmp_path):
    """weights.from_tf_variables accepts both TF naming generations and ignores optimiser slots."""
    import subprocess, sys
    from keyword_spotting_amd import get_config, weights
    cfg = get_config()
    w = weights.init_weights(cfg, seed=3)
    for new in (True, False):
        v = weights.to_tf_variables(w, new_names=new)
        v = {k + ":0": a for k, a in v.items()}
        v["model/drnn/multi_rnn_cell/cell_0/gru_cell/gates/kernel/Adam:0"] = np.zeros((168, 256), np.float32)
        back = weights.from_tf_variables(cfg, v)
        np.testing.assert_array_equal(weights.to_blob(cfg, back), weights.to_blob(cfg, w))
    bad = weights.to_tf_variables(w)
    del bad["model/drnn/multi_rnn_cell/cell_1/gru_cell/candidate/bias"]
    with pytest.raises(ValueError, match="layer 1"):
        weights.from_tf_variables(cfg, bad)
    src = tmp_path / "vars.npz"
    np.savez(src, **weights.to_tf_variables(w))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call([sys.executable, os.path.join(root, "tools", "convert_weights.py"), str(src), "--out",
                           str(tmp_path / "m")])
    np.testing.assert_array_equal(np.fromfile(tmp_path / "m.blob", np.float32), weights.to_blob(cfg, w))


def test_buf_to_float_matches_the_ring_buffer_conversion():
    """detector.py:40-43: scale = 1 / 2^(8n-1) applied to little-endian signed PCM (RingBuffer.get, :74-79)."""
    import torch
    from keyword_spotting_amd.detector import buf_to_float
    raw = np.array([-32768, -1, 0, 1, 12345, 32767], np.int16)
    want = (1.0 / float(1 << 15)) * raw.astype(np.float32)              # the reference's expression
    np.testing.assert_array_equal(buf_to_float(torch.from_numpy(raw)).numpy(), want)
    f = torch.tensor([0.25, -0.5])
    assert buf_to_float(f) is f or torch.equal(buf_to_float(f), f)      # float input passes through
    with pytest.raises(ValueError):
        buf_to_float(torch.zeros(3, dtype=torch.int64))
