"""The GRU / dense / softmax pin: outputs of the reference's own graph code under TensorFlow 1.x
(tests/golden/gru_tf_golden.npz, written by tests/golden/make_gru_golden.py on a machine that has TF 1.x).

TensorFlow is not installable in the build image, so the fixture may be absent: the comparing tests then SKIP with
"PARITY UNPINNED" (they must never pass vacuously), and only the generator's refusal path is exercised.  With the
fixture present, `-m "not gpu"` checks the oracle against TensorFlow and `-m gpu` checks the HIP path against it."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

FIXTURE = os.path.join(ROOT, "tests", "golden", "gru_tf_golden.npz")
STACK_CASES = ("A", "A5", "Achk", "B", "C", "Arelu")


def _fixture():
    if not os.path.exists(FIXTURE):
        pytest.skip("PARITY UNPINNED: %s absent (needs TensorFlow 1.x: python tests/golden/make_gru_golden.py)"
                    % os.path.relpath(FIXTURE, ROOT))
    return np.load(FIXTURE)


def _case(z, tag):
    from keyword_spotting_amd import get_config, weights
    n_mel, hidden, layers, classes, relu, clip = [int(v) for v in z[tag + "/shape"]]
    cfg = get_config(n_mel=n_mel, hidden_size=hidden, num_layers=layers, use_relu=bool(relu), value_clip=float(clip))
    assert cfg.num_classes == classes
    prefix = tag + "/var/"
    w = weights.from_tf_variables(cfg, {k[len(prefix):]: z[k] for k in z.files if k.startswith(prefix)})
    return cfg, w


def test_generator_refuses_loudly_without_tensorflow(tmp_path):
    """No TF here: the script must say so and exit 2 without writing anything (and must not stub anything)."""
    try:
        import tensorflow  # noqa: F401
        pytest.skip("tensorflow is importable here: run the generator instead")
    except ImportError:
        pass
    out = tmp_path / "x.npz"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "golden", "make_gru_golden.py"), "--out", str(out)],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 2, (r.returncode, r.stderr)
    assert "UNPINNED" in r.stderr or "not found" in r.stderr
    assert not out.exists()


@pytest.mark.parametrize("tag", STACK_CASES)
def test_oracle_matches_tensorflow(tag):
    from oracle import gru_oracle as G
    z = _fixture()
    cfg, w = _case(z, tag)
    kw = dict(use_relu=cfg.use_relu, value_clip=cfg.value_clip)
    if tag + "/chunks" in z.files:
        state, outs, pos = z[tag + "/state0"].astype(np.float64), [], 0
        for n in z[tag + "/chunks"]:
            lg, state = G.gru_forward(w, z[tag + "/mel"][:, pos:pos + n], state, dtype=np.float64, **kw)
            outs.append(lg); pos += int(n)
        lg = np.concatenate(outs, 1)
    else:
        lg, state = G.gru_forward(w, z[tag + "/mel"], z[tag + "/state0"], seq_len=z[tag + "/seq_len"], dtype=np.float64, **kw)
    # TF computes in fp32 (Eigen); the fp64 oracle differs from it by fp32 round-off only
    assert np.abs(lg - z[tag + "/logits"]).max() < 2e-5
    assert np.abs(state - z[tag + "/state"]).max() < 2e-5
    assert np.abs(G.softmax(lg) - z[tag + "/softmax"]).max() < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("tag", STACK_CASES)
def test_hip_path_matches_tensorflow(tag):
    import torch
    from keyword_spotting_amd.rnn_ctc import DeployModel
    z = _fixture()
    cfg, w = _case(z, tag)
    m = DeployModel(cfg, w)
    mel = torch.from_numpy(z[tag + "/mel"]).cuda()
    if tag + "/chunks" in z.files:
        state, lgs, sms, pos = torch.from_numpy(z[tag + "/state0"]).cuda(), [], [], 0
        for n in z[tag + "/chunks"]:
            r = m.forward(mel[:, pos:pos + int(n)].contiguous(), state)
            state = r["state"]; lgs.append(r["logits"]); sms.append(r["softmax"]); pos += int(n)
        lg, sm = torch.cat(lgs, 1), torch.cat(sms, 1)
    else:
        r = m.forward(mel, torch.from_numpy(z[tag + "/state0"]), seq_len=torch.from_numpy(z[tag + "/seq_len"]))
        lg, sm, state = r["logits"], r["softmax"], r["state"]
    assert np.abs(lg.cpu().numpy() - z[tag + "/logits"]).max() < 1e-4          # north_star tolerance
    assert np.abs(state.cpu().numpy() - z[tag + "/state"]).max() < 1e-4
    assert np.abs(sm.cpu().numpy() - z[tag + "/softmax"]).max() < 2e-5


@pytest.mark.gpu
def test_hip_deploy_graph_with_pcm_feed_matches_tensorflow():
    """Case D: the shipped DeployModel graph, PCM in (pins tf_frame, rfft and the librosa mel basis as well)."""
    import torch
    from keyword_spotting_amd.rnn_ctc import DeployModel
    z = _fixture()
    if "D/shape" not in z.files:
        pytest.skip("PARITY UNPINNED for the front-end: the fixture was generated without the DeployModel case")
    cfg, w = _case(z, "D")
    m = DeployModel(cfg, w)
    state = torch.zeros(cfg.num_layers, 1, cfg.hidden_size, device="cuda")
    for c in range(4):
        sm, lg, state = m.run(["model/softmax:0", "model/logit:0", "model/rnn_states:0"],
                              {"model/inputX:0": z["D/chunk%d/data" % c], "model/rnn_initial_states:0": state})
        want_l = z["D/chunk%d/logit" % c]
        scale = max(1.0, float(np.abs(want_l).max()))
        assert np.abs(lg.cpu().numpy() - want_l).max() < 1e-4 * scale
        assert np.abs(sm.cpu().numpy() - z["D/chunk%d/softmax" % c].reshape(sm.shape)).max() < 2e-5
        assert np.abs(state.cpu().numpy() - z["D/chunk%d/state" % c]).max() < 1e-4
