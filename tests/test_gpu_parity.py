"""HIP GRU path vs the oracle, through the C ABI (kws_step).  Tolerance from north_star: logits
within 1e-4, identical collapsed keyword decisions (margin-aware at the hard thresholds)."""
import numpy as np
import pytest
import torch

from oracle import decode_oracle as D
from oracle import gru_oracle as G

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _model(shape, w, kernel="auto", **cfg_kw):
    from keyword_spotting_amd import get_config
    from keyword_spotting_amd.rnn_ctc import DeployModel
    i, h, l, c = shape
    if kernel == "f16x3":        # the split-fp16 path has to meet every fp32 tolerance: it rides along as a "kernel" here
        kernel, cfg_kw = "auto", dict(cfg_kw, precision="f16x3")
    cfg = get_config(n_mel=i, hidden_size=h, num_layers=l, **cfg_kw)
    if c != 6:
        cfg.label_dict = {str(k): k for k in range(1, c - 2)}
    return DeployModel(cfg, w, kernel=kernel)


def _cfg_tuple(shape, use_relu=0, clip=-1.0):
    return shape + (use_relu, clip)


SHAPES = {"A": (40, 128, 2, 6), "B": (60, 128, 2, 6), "C": (60, 256, 4, 6),
          "D": (64, 128, 2, 6), "E": (32, 128, 1, 6), "F": (48, 128, 3, 6),     # the other resident front-end widths
          "G": (12, 128, 4, 5),                                                 # f16x3 only: narrow input, four layers, C = 5
          "H": (24, 256, 2, 6)}                                                 # f16x3 streaming kernels: one (padded) input chunk, two layers


@pytest.mark.parametrize("name,kernel,batch,frames", [
    ("A", "resident", 37, 50), ("A", "generic", 37, 50), ("A", "resident", 1, 300),
    ("A", "resident", 16, 1), ("A", "resident", 33, 7), ("B", "resident", 20, 23), ("B", "generic", 5, 22),
    ("C", "generic", 20, 12), ("C", "auto", 3, 40),
    ("D", "resident", 19, 30), ("E", "resident", 35, 26), ("F", "resident", 8, 41),
    ("A", "f16x3", 37, 50), ("A", "f16x3", 1, 300), ("A", "f16x3", 16, 1), ("A", "f16x3", 33, 7), ("B", "f16x3", 20, 23),
    ("D", "f16x3", 19, 30), ("E", "f16x3", 35, 26), ("F", "f16x3", 8, 41), ("G", "f16x3", 21, 19),
    # hidden = 256: the L2-streaming f16x3 kernels (gru_f16x3_generic.hip), layer-pipelined at these sizes
    ("C", "f16x3", 20, 12), ("C", "f16x3", 3, 40), ("C", "f16x3", 1, 1), ("H", "f16x3", 37, 9)])
def test_logits_state_softmax_match_oracle(oracle_c, name, kernel, batch, frames):
    shape = SHAPES[name]
    i, h, l, c = shape
    w = G.random_weights(i, h, l, c, seed=21)
    mel = G.synthetic_mel(batch, frames, i, seed=22)
    st0 = (0.5 * np.random.default_rng(23).standard_normal((l, batch, h))).astype(np.float32)
    want_l, want_s = G.gru_forward(w, mel, st0, dtype=np.float64)
    m = _model(shape, w, kernel)
    r = m.forward(torch.from_numpy(mel), torch.from_numpy(st0))
    got_l, got_sm, got_s = r["logits"].cpu().numpy(), r["softmax"].cpu().numpy(), r["state"].cpu().numpy()
    assert np.abs(got_l - want_l).max() < TOL
    assert np.abs(got_s - want_s).max() < TOL
    assert np.abs(got_sm - G.softmax(want_l)).max() < 2e-5
    np.testing.assert_allclose(got_sm.sum(-1), 1.0, atol=1e-6)
    # and against the C restatement (what bench.py times as the CPU baseline)
    c_l, _, c_s = oracle_c.gru_forward(_cfg_tuple(shape), G.weights_to_blob(w), mel, st0)
    assert np.abs(got_l - c_l).max() < TOL and np.abs(got_s - c_s).max() < TOL


def test_resident_and_generic_kernels_agree():
    w = G.random_weights(40, 128, 2, 6, seed=31)
    mel = torch.from_numpy(G.synthetic_mel(48, 64, 40, seed=32))
    a = _model(SHAPES["A"], w, "resident")
    b = _model(SHAPES["A"], w, "generic")
    ra, rb = a.forward(mel, a.zero_state(48)), b.forward(mel, b.zero_state(48))
    assert (ra["logits"] - rb["logits"]).abs().max().item() < 2e-5   # same math, different K order
    assert (ra["state"] - rb["state"]).abs().max().item() < 2e-5


@pytest.mark.parametrize("kernel", ["resident", "generic", "f16x3"])
def test_chunked_streaming_equals_one_shot_bitwise(kernel):
    """detector.py:254-289 test2: feeding 21/22/23-frame chunks with the carried state must give the
    single-call result -- bit-exact on the GPU, every frame runs the identical instruction sequence."""
    w = G.init_weights()
    mel = torch.from_numpy(G.synthetic_mel(19, 300)).cuda()
    m = _model(SHAPES["A"], w, kernel)
    whole = m.forward(mel, m.zero_state(19))
    state, outs, pos = m.zero_state(19), [], 0
    for n in D.chunk_frame_counts([3600] * 13) + [300 - sum(D.chunk_frame_counts([3600] * 13))]:
        lg, state = m.step(mel[:, pos:pos + n].contiguous(), state)
        outs.append(lg)
        pos += n
    assert pos == 300
    assert torch.equal(torch.cat(outs, 1), whole["logits"])
    assert torch.equal(state, whole["state"])


@pytest.mark.parametrize("kernel", ["auto", "f16x3"])
def test_state_may_alias_and_empty_calls(kernel):
    w = G.init_weights()
    m = _model(SHAPES["A"], w, kernel)
    mel = torch.from_numpy(G.synthetic_mel(4, 9)).cuda()
    st = (0.1 * torch.randn(2, 4, 128, device="cuda"))
    ref = m.forward(mel, st.clone())
    st2 = st.clone()
    r = m.forward(mel, st2, state_out=st2)                       # in place
    assert torch.equal(r["state"], ref["state"]) and r["state"].data_ptr() == st2.data_ptr()
    e = m.forward(mel[:, :0].contiguous(), st)                   # T == 0: state passes through
    assert torch.equal(e["state"], st) and e["logits"].shape == (4, 0, 6)
    # ... except for the streams clean_state() (detector.py:313-316) zeroed before this sess.run over zero frames
    mask = torch.tensor([0, 1, 0, 1], dtype=torch.uint8)
    e = m.forward(mel[:, :0].contiguous(), st, reset_mask=mask)
    assert torch.equal(e["state"][:, 0::2], st[:, 0::2]) and not e["state"][:, 1::2].any()
    st3 = st.clone()
    m.forward(mel[:, :0].contiguous(), st3, reset_mask=mask, state_out=st3)          # in place
    assert torch.equal(st3, e["state"])
    # ... and their ctc_decode2 carry restarts from pre_word = -1, as a reset does in a call with frames
    pw = torch.tensor([2, 2, 0, 1], dtype=torch.int32, device="cuda")
    m.forward(mel[:, :0].contiguous(), st, reset_mask=mask, prev_word=pw)
    assert pw.tolist() == [2, -1, 0, -1]
    e = m.forward(mel[:0].contiguous(), st[:, :0].contiguous())  # B == 0
    assert e["logits"].shape == (0, 9, 6)


@pytest.mark.parametrize("kernel", ["resident", "generic", "f16x3"])
def test_sequence_length_and_reset_mask(kernel):
    w = G.random_weights(40, 128, 2, 6, seed=41)
    b, t = 21, 30
    mel = G.synthetic_mel(b, t, 40, seed=42)
    rng = np.random.default_rng(43)
    st0 = (0.5 * rng.standard_normal((2, b, 128))).astype(np.float32)
    lens = rng.integers(0, t + 1, b).astype(np.int32)
    lens[:3] = [0, t, 1]
    reset = (rng.random(b) < 0.4).astype(np.uint8)
    st_ref = st0 * (1 - reset)[None, :, None]
    want_l, want_s = G.gru_forward(w, mel, st_ref, seq_len=lens, dtype=np.float64)
    m = _model(SHAPES["A"], w, kernel)
    r = m.forward(torch.from_numpy(mel), torch.from_numpy(st0), seq_len=torch.from_numpy(lens),
                  reset_mask=torch.from_numpy(reset))
    assert np.abs(r["logits"].cpu().numpy() - want_l).max() < TOL
    assert np.abs(r["state"].cpu().numpy() - want_s).max() < TOL
    # rows past seq_len are exactly the bias row (zero output of dynamic_rnn)
    got = r["logits"].cpu().numpy()
    for k in range(b):
        np.testing.assert_array_equal(got[k, lens[k]:], np.broadcast_to(w["bfc"], (t - lens[k], 6)))


@pytest.mark.parametrize("kernel", ["auto", "f16x3"])
def test_relu_and_clip(kernel):
    w = G.random_weights(40, 128, 2, 6, seed=51)
    w["Wfc"] *= 20
    mel = G.synthetic_mel(6, 25, 40, seed=52)
    want, _ = G.gru_forward(w, mel, use_relu=True, value_clip=1.0, dtype=np.float64)
    m = _model(SHAPES["A"], w, kernel, use_relu=True, value_clip=1.0)
    got = m.forward(torch.from_numpy(mel), m.zero_state(6))["logits"].cpu().numpy()
    assert np.abs(got - want).max() < 2e-3 * 20 / 20 + 5e-4      # logits up to 20: relative 1e-4-class
    assert got.min() == 0.0 and got.max() == 20.0


def _margin_ok(softmax_row, thres, eps=1e-4):
    p = softmax_row[1:5]
    srt = np.sort(p)
    return abs(p.max() - thres) > eps and (srt[-1] - srt[-2]) > eps


@pytest.mark.parametrize("kernel", ["resident", "generic", "f16x3"])
def test_fused_decode2_tokens_and_carry(kernel):
    """Fused per-frame ctc_decode2 events == reference rule on the oracle's softmax; pre_word is
    carried across chunk boundaries (== decoding the concatenation)."""
    w = G.init_weights()
    b, t = 40, 120
    mel = G.synthetic_mel(b, t, 40, seed=61)
    want_l, _ = G.gru_forward(w, mel, dtype=np.float64)
    want_sm = G.softmax(want_l)
    m = _model(SHAPES["A"], w, kernel)
    state, pw = m.zero_state(b), m.fresh_prev_word(b)
    toks, pos = [], 0
    melc = torch.from_numpy(mel).cuda()
    for n in (21, 22, 23, 22, 1, 31):
        r = m.forward(melc[:, pos:pos + n].contiguous(), state, prev_word=pw)
        state = r["state"]
        toks.append(r["tokens"])
        pos += n
    assert pos == t
    toks = torch.cat(toks, 1).cpu().numpy()
    from keyword_spotting_amd.prediction import tokens_to_seq
    checked = 0
    for k in range(b):
        if not all(_margin_ok(row, 0.4) for row in want_sm[k]):
            continue                                   # a frame sits within 1e-4 of the threshold / a tie
        np.testing.assert_array_equal(tokens_to_seq(toks[k]), D.ctc_decode2(want_sm[k], 6), err_msg=str(k))
        checked += 1
    assert checked >= b // 2
    assert (toks > 0).sum() > b                       # the decode is not vacuous
    # carried pre_word == word of the last frame
    last = D.frame_words(want_sm[:, -1, :].reshape(b, 6), 1, 5, 0.4)
    ok = [k for k in range(b) if _margin_ok(want_sm[k, -1], 0.4)]
    np.testing.assert_array_equal(pw.cpu().numpy()[ok], last[ok])


def test_errors_are_reported_not_fatal():
    from keyword_spotting_amd import _lib, get_config
    from keyword_spotting_amd.rnn_ctc import DeployModel
    w = G.init_weights()
    m = _model(SHAPES["A"], w)
    with pytest.raises(_lib.InvalidArgumentError):
        m.forward(torch.zeros(2, 5, 41), m.zero_state(2))          # wrong n_mel
    with pytest.raises(_lib.InvalidArgumentError):
        m.forward(torch.zeros(2, 5, 40), m.zero_state(3))          # state batch mismatch
    with pytest.raises(_lib.InvalidArgumentError):
        m.run(["model/nope:0"], {"model/inputX:0": torch.zeros(5, 40), "model/rnn_initial_states:0": m.zero_state(1)})
    with pytest.raises(_lib.UnsupportedError):
        DeployModel(get_config(hidden_size=100), np.zeros(10, np.float32))
    with pytest.raises(_lib.InvalidArgumentError):
        DeployModel(get_config(), np.zeros(10, np.float32))        # blob size
    with pytest.raises(_lib.UnsupportedError):
        _model((50, 128, 2, 6), G.init_weights(50), "resident")


def test_session_run_surface_batch1():
    """detector.py:190-193 vocabulary at batch 1: 2-D softmax [T,C], 3-D logit [1,T,C], state [L,1,H]."""
    w = G.init_weights()
    m = _model(SHAPES["A"], w)
    mel = G.synthetic_mel(1, 22)
    softmax, state = m.run(["model/softmax:0", "model/rnn_states:0"],
                           {"model/inputX:0": mel[0], "model/rnn_initial_states:0": np.zeros((2, 1, 128), np.float32)})
    assert tuple(softmax.shape) == (22, 6) and tuple(state.shape) == (2, 1, 128)
    logit = m.run("model/logit:0", {"model/inputX:0": mel[0], "model/rnn_initial_states:0": m.zero_state(1)})
    assert tuple(logit.shape) == (1, 22, 6)
    want_l, want_s = G.gru_forward(w, mel, dtype=np.float64)
    assert np.abs(logit.cpu().numpy() - want_l).max() < TOL
    assert np.abs(softmax.cpu().numpy() - G.softmax(want_l)[0]).max() < 2e-5


@pytest.mark.parametrize("kernel", ["auto", "f16x3"])
def test_full_size_properties(oracle_c, kernel):
    """BASELINE config 2 at full size (B=4096, T=300): (i) a sample of streams against the C oracle,
    (ii) batch-composition independence: a stream's result does not depend on its batch neighbours
    (what multi-GPU sharding relies on), (iii) chunked == one-shot, bitwise."""
    w = G.init_weights()
    b, t = 4096, 300
    rng = torch.Generator(device="cpu").manual_seed(71)
    mel = (torch.randn(b, t, 40, generator=rng).abs() * 2).cuda()
    m = _model(SHAPES["A"], w, kernel)
    whole = m.forward(mel, m.zero_state(b))
    pick = [0, 1, 15, 16, 17, 2047, 2048, 4079, 4080, 4095] + list(range(100, 4000, 211))
    sub = mel[pick].cpu().numpy()
    c_l, _, c_s = oracle_c.gru_forward(_cfg_tuple(SHAPES["A"]), G.weights_to_blob(w), sub,
                                       np.zeros((2, len(pick), 128), np.float32), threads=4)
    assert np.abs(whole["logits"][pick].cpu().numpy() - c_l).max() < TOL
    assert np.abs(whole["state"][:, pick].cpu().numpy() - c_s).max() < TOL
    part = m.forward(mel[pick].contiguous(), m.zero_state(len(pick)))
    assert torch.equal(part["logits"], whole["logits"][pick])
    assert torch.equal(part["state"], whole["state"][:, pick])
    state, pos = m.zero_state(b), 0
    for n in (150, 22, 128):
        lg, state = m.step(mel[:, pos:pos + n].contiguous(), state)
        assert torch.equal(lg, whole["logits"][:, pos:pos + n])
        pos += n
    assert torch.equal(state, whole["state"])


@pytest.mark.parametrize("kernel,precision", [("resident", "fp32"), ("generic", "fp32"), ("auto", "bf16"), ("auto", "f16x3")])
def test_saturating_inputs_stay_finite_and_match_the_oracle(kernel, precision):
    """Mel frames of magnitude 1e3..1e4 drive every gate deep into saturation: exp2 overflows to inf and the reciprocal
    returns 0 -- sigmoid/tanh must land exactly on 0 / 1 / -1 like the oracle's, never on NaN."""
    w = G.init_weights(seed=11)
    b, t = 21, 40
    mel = G.synthetic_mel(b, t, 40, seed=12) * np.float32(2e3)
    mel[3] *= 0                                                       # an all-zero stream beside them
    st0 = (0.9 * np.random.default_rng(13).standard_normal((2, b, 128))).clip(-1, 1).astype(np.float32)
    m = _model(SHAPES["A"], w, kernel, precision=precision)
    r = m.forward(torch.from_numpy(mel), torch.from_numpy(st0))
    got_l, got_s = r["logits"].cpu().numpy(), r["state"].cpu().numpy()
    assert np.isfinite(got_l).all() and np.isfinite(got_s).all() and np.isfinite(r["softmax"].cpu().numpy()).all()
    if precision in ("fp32", "f16x3"):
        want_l, want_s = G.gru_forward(w, mel, st0, dtype=np.float64)
        tol_l, tol_s = 5e-4, 1e-4        # layer-0 pre-activations ~1e4: fp32 products carry ~1e-3 absolute, gates are saturated
    else:
        want_l, want_s = G.gru_forward_bf16(w, mel, st0)
        tol_l, tol_s = 8e-2, 3e-2
    assert np.abs(got_s - want_s).max() < tol_s and np.abs(got_l - want_l).max() < tol_l
    np.testing.assert_allclose(r["softmax"].cpu().numpy().sum(-1), 1.0, atol=1e-5)


@pytest.mark.parametrize("scale", [1e-6, 1e-3, 1.0, 3e2, 1e5, 1e6])
def test_f16x3_covers_the_mel_dynamic_range(scale):
    """The split path feeds the matrix pipe fp16 pieces of mel * 2^-8: magnitudes from 1e-6 to 1e7 (fp16 alone ends at 65504; the largest sample here is ~9e6)
    must still meet the fp32 tolerance -- weights scaled down so that the gates do not simply saturate; above 1.6e7 the input
    saturates (finite results, documented)."""
    w = G.random_weights(40, 128, 2, 6, seed=81)
    for lay in w["layers"][:1]:
        lay["Wg"][:40] *= np.float32(min(1.0, 1.0 / scale))
        lay["Wc"][:40] *= np.float32(min(1.0, 1.0 / scale))
    b, t = 18, 24
    mel = (G.synthetic_mel(b, t, 40, seed=82) * np.float32(scale)).astype(np.float32)
    want_l, want_s = G.gru_forward(w, mel, dtype=np.float64)
    m = _model(SHAPES["A"], w, "f16x3")
    r = m.forward(torch.from_numpy(mel), m.zero_state(b))
    assert np.abs(r["logits"].cpu().numpy() - want_l).max() < TOL
    assert np.abs(r["state"].cpu().numpy() - want_s).max() < TOL
    huge = m.forward(torch.from_numpy(mel * np.float32(1e30 / scale)), m.zero_state(b))
    assert torch.isfinite(huge["logits"]).all() and torch.isfinite(huge["state"]).all()


def test_f16x3_preconditions_and_its_distance_from_the_fp32_kernels():
    from keyword_spotting_amd import _lib
    w = G.init_weights()
    big = {k: v for k, v in w.items()}
    big["layers"] = [dict(lay) for lay in w["layers"]]
    big["layers"][1]["Wc"] = big["layers"][1]["Wc"].copy()
    big["layers"][1]["Wc"][3, 5] = 200.0
    with pytest.raises(_lib.UnsupportedError):
        _model(SHAPES["A"], big, "f16x3")                         # |w| >= 64 is outside the fp16 pieces' range (kws_amd.h)
    ok = {k: v for k, v in w.items()}
    ok["layers"] = [dict(lay) for lay in w["layers"]]
    ok["layers"][0]["bg"] = ok["layers"][0]["bg"].copy()
    ok["layers"][0]["bg"][7] = 100.0                              # biases (and bfc) stay fp32 in the kernels: any magnitude is fine
    ok["bfc"] = ok["bfc"].copy()
    ok["bfc"][2] = -300.0
    _model(SHAPES["A"], ok, "f16x3")
    with pytest.raises(_lib.UnsupportedError):
        _model((40, 64, 2, 6), G.init_weights(40, 64, 2, 6), "f16x3")   # hidden 64: neither the resident (128) nor the streaming (256) kernels
    mel = torch.from_numpy(G.synthetic_mel(64, 300, 40, seed=83))
    a, f = _model(SHAPES["A"], w, "resident"), _model(SHAPES["A"], w, "f16x3")
    ra = a.forward(mel, a.zero_state(64), prev_word=a.fresh_prev_word(64))
    rf = f.forward(mel, f.zero_state(64), prev_word=f.fresh_prev_word(64))
    d = (ra["logits"] - rf["logits"]).abs().max().item()
    print("f16x3 vs fp32 resident kernels, 64 streams x 300 frames: max |dlogit| %.2e, token frames differing %d of %d"
          % (d, int((ra["tokens"] != rf["tokens"]).sum()), ra["tokens"].numel()))
    assert d < 2e-5                                               # not bit-identical (another summation order), fp32-rounding close
    assert f.kernel_names() == ["gru_layer_f16x3<2, true, false>", "gru_layer_f16x3<4, false, true>"]


def test_f16x3_streaming_kernels_masks_chunks_and_launch_layouts():
    """hidden = 256 at fp32 tolerance on the fp16 matrix pipe (gru_f16x3_generic.hip): sequence lengths and reset masks as
    dynamic_rnn / clean_state define them, chunked calls == one call bit for bit, and the two launch layouts -- all layers in
    one layer-pipelined grid (L x groups <= CUs) and one launch per layer -- agree bit for bit on shared streams."""
    shape = SHAPES["C"]
    i, h, l, c = shape
    w = G.random_weights(i, h, l, c, seed=91)
    b, t = 23, 30
    rng = np.random.default_rng(92)
    mel = G.synthetic_mel(b, t, i, seed=93)
    st0 = (0.5 * rng.standard_normal((l, b, h))).astype(np.float32)
    lens = rng.integers(0, t + 1, b).astype(np.int32)
    lens[0], lens[1] = 0, t
    reset = (rng.random(b) < 0.3).astype(np.uint8)
    st_eff = st0.copy()
    st_eff[:, reset != 0] = 0
    want_l, want_s = G.gru_forward(w, mel, st_eff, seq_len=lens, dtype=np.float64)
    m = _model(shape, w, "f16x3")
    r = m.forward(torch.from_numpy(mel), torch.from_numpy(st0), seq_len=torch.from_numpy(lens), reset_mask=torch.from_numpy(reset),
                  prev_word=m.fresh_prev_word(b))
    assert m.kernel_names()[-1].startswith("gru_stack_f16x3_pipelined<4>")
    got_l, got_s = r["logits"].cpu().numpy(), r["state"].cpu().numpy()
    assert np.abs(got_l - want_l).max() < TOL and np.abs(got_s - want_s).max() < TOL
    bias_row = r["logits"][0, 0].clone()
    for k in range(b):                                         # the zero-output row past seq_len: logits = bfc, bit for bit
        if lens[k] < t:
            assert torch.equal(r["logits"][k, int(lens[k]):], bias_row.expand(t - int(lens[k]), -1))
    # chunked == one shot
    x = torch.from_numpy(mel).cuda()
    whole = m.forward(x, torch.from_numpy(st0))
    state, pos, parts = torch.from_numpy(st0).cuda(), 0, []
    for n in (7, 1, 16, 6):
        rr = m.forward(x[:, pos:pos + n].contiguous(), state)
        parts.append(rr["logits"]); state = rr["state"]; pos += n
    assert torch.equal(torch.cat(parts, 1), whole["logits"]) and torch.equal(state, whole["state"])
    # more groups than the pipelined grid takes (L x groups > CUs): one launch per layer, same bits on the shared streams
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    big_b = 16 * (cus // l + 1)
    big = torch.from_numpy(G.synthetic_mel(big_b, 6, i, seed=94)).cuda()
    seq = m.forward(big, m.zero_state(big_b))
    assert m.kernel_names() == ["gru_layer_f16x3_generic<4, true, false>"] + ["gru_layer_f16x3_generic<4, false, false>"] * (l - 2) + \
        ["gru_layer_f16x3_generic<4, false, true>"]
    pipe = m.forward(big[:48].contiguous(), m.zero_state(48))
    assert torch.equal(pipe["logits"], seq["logits"][:48]) and torch.equal(pipe["state"], seq["state"][:, :48])
    m.status()


def test_f16x3_streaming_kernels_epilogue_relu_clip_tokens_and_five_classes():
    """The hidden = 256 f16x3 kernels share the last-layer epilogue of every other family; its switches through THIS launch:
    relu + clip (models/rnn_ctc.py:280-283), the fused ctc_decode2 tokens with their carry across calls, C = 5 classes, a T == 0
    call with a reset mask, state in place."""
    i, h, l = 60, 256, 2
    # relu + clip
    w = G.random_weights(i, h, l, 6, seed=95)
    w["Wfc"] *= 20
    mel = G.synthetic_mel(7, 19, i, seed=96)
    want, _ = G.gru_forward(w, mel, use_relu=True, value_clip=1.0, dtype=np.float64)
    m = _model((i, h, l, 6), w, "f16x3", use_relu=True, value_clip=1.0)
    got = m.forward(torch.from_numpy(mel), m.zero_state(7))["logits"].cpu().numpy()
    assert np.abs(got - want).max() < 2.5e-3 and got.min() == 0.0 and got.max() == 20.0
    # tokens + carry across calls == one call; == ctc_decode2 of the oracle's softmax wherever no frame sits on a threshold / tie
    w = G.random_weights(i, h, l, 6, seed=97)
    w["Wfc"] = (w["Wfc"] * 0.5).astype(np.float32)          # (h = 256 saturates the softmax quickly: a soft projection keeps words coming)
    b, t = 20, 48
    mel = G.synthetic_mel(b, t, i, seed=98)
    m = _model((i, h, l, 6), w, "f16x3")
    x = torch.from_numpy(mel).cuda()
    pw = m.fresh_prev_word(b)
    whole = m.forward(x, m.zero_state(b), prev_word=pw)
    pw2, st, toks = m.fresh_prev_word(b), m.zero_state(b), []
    for lo, hi in ((0, 17), (17, 18), (18, 48)):
        r = m.forward(x[:, lo:hi].contiguous(), st, prev_word=pw2, state_out=st)
        toks.append(r["tokens"])
    assert torch.equal(torch.cat(toks, 1), whole["tokens"]) and torch.equal(pw2, pw) and torch.equal(st, whole["state"])
    want_l, _ = G.gru_forward(w, mel, dtype=np.float64)
    sm = G.softmax(want_l)
    from keyword_spotting_amd.prediction import tokens_to_seq
    checked = 0
    for k in range(b):
        # (a tie among the word classes only matters where the best of them clears the threshold)
        if all(abs(sm[k, f, 1:5].max() - 0.4) > 1e-4 and (sm[k, f, 1:5].max() < 0.4 or _margin_ok(sm[k, f], 0.4)) for f in range(t)):
            assert np.array_equal(tokens_to_seq(whole["tokens"][k].cpu().numpy()), D.ctc_decode2(sm[k], 6)), k
            checked += 1
    assert checked >= 3 and int((whole["tokens"] > 0).sum()) > 0
    # zero frames with a reset mask; state in place
    mask = torch.tensor([1, 0] * (b // 2), dtype=torch.uint8)
    st2 = st.clone()
    m.forward(x[:, :0].contiguous(), st2, reset_mask=mask, state_out=st2)
    assert not st2[:, 0::2].any() and torch.equal(st2[:, 1::2], st[:, 1::2])
    # five classes
    w5 = G.random_weights(i, h, l, 5, seed=99)
    mel5 = G.synthetic_mel(5, 11, i, seed=100)
    want5, want5s = G.gru_forward(w5, mel5, dtype=np.float64)
    m5 = _model((i, h, l, 5), w5, "f16x3")
    r5 = m5.forward(torch.from_numpy(mel5), m5.zero_state(5))
    assert np.abs(r5["logits"].cpu().numpy() - want5).max() < TOL and np.abs(r5["state"].cpu().numpy() - want5s).max() < TOL
    assert np.abs(r5["softmax"].cpu().numpy() - G.softmax(want5)).max() < 2e-5
