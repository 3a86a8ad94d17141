"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy) of the reference's streaming GRU path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product (keyword_spotting_amd/) never imports anything under oracle/.

PARITY PARTIAL for this file (third-party pins only, see the end of this paragraph): the reference's GRU / dense / softmax arithmetic lives in
TensorFlow 1.x (`tensorflow.contrib.rnn.GRUCell`, `MultiRNNCell`, `dynamic_rnn`), a third-party
dependency that is neither vendored in /root/reference nor version-pinned (no requirements file;
era: TF 1.1 - 1.3, see utils/stft.py:26, models/rnn_ctc.py:182) and cannot be installed here.
The reference holds no test or golden vector for this stage.  This module therefore restates the
published TF-1.x GRUCell algorithm and is anchored on the reference's own call sites; what pins it from
outside is TensorFlow's own published unit-test constants for GRUCell / MultiRNNCell
(tests/test_tf_published_kat.py: 0.175991, 0.156736, 0.13248) and torch.nn.GRU over whole sequences on the weight
family where the two cells coincide (tests/test_torch_gru_pin.py) -- third-party evidence, not the reference
run here, hence "partial" until tests/golden/make_gru_golden.py has run under TF 1.x:

  models/rnn_ctc.py:179-199  get_cell      -> plain GRUCell(num_units=H, activation=tanh)
  models/rnn_ctc.py:202-244  inference1    -> MultiRNNCell + dynamic_rnn, batch-major, initial_state tuple
  models/rnn_ctc.py:247-284  inference2    -> logits = flat(h) @ W[H,C] + b (+ optional relu / clip[0,20])
  models/rnn_ctc.py:156-165  DeployModel   -> unstack state, stack state, logit, softmax

TF-1.x GRUCell.call (restated):
  g    = sigmoid([x, h] @ Wg + bg)          Wg:[I+H, 2H]  bg:[2H] (initialised to 1.0)
  r, u = split(g, 2, axis=1)                r first
  c    = tanh([x, r*h] @ Wc + bc)           Wc:[I+H, H]   bc:[H]
  h'   = u*h + (1-u)*c                      output == new state
TF dynamic_rnn with sequence_length: for t >= seq_len[b] the emitted output row is zero and the
whole state tuple of row b is copied through unchanged.

Three independent formulations live here and are cross-checked by tests/test_oracle_gru.py:
  * gru_forward(dtype=float32|float64)     -- concatenated-matmul form (as TF writes it)
  * gru_forward_split                      -- split-weight form (x-part / h-part summed separately)
  * oracle/torch_eager.py                  -- op-by-op torch CPU form (also the "TF-CPU stand-in" timing)
"""
import numpy as np


# --------------------------------------------------------------------------------------
# weights: canonical layout == TF variable layout
#   per layer l: Wg [I_l+H, 2H], bg [2H], Wc [I_l+H, H], bc [H]   (I_0 = n_mel, I_l = H)
#   Wfc [H, C], bfc [C]
# --------------------------------------------------------------------------------------
def init_weights(n_mel=40, hidden=128, num_layers=2, num_classes=6, seed=0):
    """SURVEY.md 8(d) config-1 initialisation: glorot-uniform kernels (TF default for GRUCell's
    _linear), gate bias 1.0, candidate bias 0 (TF GRUCell), fc ~ truncated N(0,1) at 2 sigma
    (models/rnn_ctc.py:265-268), fc bias 0 (:271-273)."""
    rng = np.random.default_rng(seed)
    layers = []
    for l in range(num_layers):
        i_l = n_mel if l == 0 else hidden
        k = i_l + hidden
        a = np.sqrt(6.0 / (k + 2 * hidden))
        wg = rng.uniform(-a, a, size=(k, 2 * hidden)).astype(np.float32)
        a = np.sqrt(6.0 / (k + hidden))
        wc = rng.uniform(-a, a, size=(k, hidden)).astype(np.float32)
        layers.append(dict(Wg=wg, bg=np.ones(2 * hidden, np.float32),
                           Wc=wc, bc=np.zeros(hidden, np.float32)))
    wfc = rng.standard_normal((hidden, num_classes))
    bad = np.abs(wfc) > 2.0
    while bad.any():
        wfc[bad] = rng.standard_normal(int(bad.sum()))
        bad = np.abs(wfc) > 2.0
    return dict(layers=layers, Wfc=wfc.astype(np.float32),
                bfc=np.zeros(num_classes, np.float32))


def random_weights(n_mel, hidden, num_layers, num_classes, seed):
    """Fully random variant (non-trivial biases) so tests exercise every term."""
    rng = np.random.default_rng(seed)
    w = init_weights(n_mel, hidden, num_layers, num_classes, seed)
    for lay in w["layers"]:
        lay["bg"] = (1.0 + 0.3 * rng.standard_normal(2 * hidden)).astype(np.float32)
        lay["bc"] = (0.3 * rng.standard_normal(hidden)).astype(np.float32)
    w["bfc"] = (0.5 * rng.standard_normal(num_classes)).astype(np.float32)
    return w


def weights_to_blob(w):
    """Flat fp32 blob in canonical order (the C-ABI's weights_blob)."""
    parts = []
    for lay in w["layers"]:
        parts += [lay["Wg"].ravel(), lay["bg"].ravel(), lay["Wc"].ravel(), lay["bc"].ravel()]
    parts += [w["Wfc"].ravel(), w["bfc"].ravel()]
    return np.ascontiguousarray(np.concatenate(parts).astype(np.float32))


def synthetic_mel(batch, frames, n_mel=40, seed=1):
    """|N(0,1)|*2, non-negative like a magnitude mel."""
    rng = np.random.default_rng(seed)
    return (np.abs(rng.standard_normal((batch, frames, n_mel))) * 2.0).astype(np.float32)


# --------------------------------------------------------------------------------------
def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def gru_cell(x, h, lay, dtype):
    """One TF-1.x GRUCell step, concatenated form.  x:[B,I] h:[B,H]."""
    hdim = h.shape[1]
    wg, bg = lay["Wg"].astype(dtype), lay["bg"].astype(dtype)
    wc, bc = lay["Wc"].astype(dtype), lay["bc"].astype(dtype)
    g = _sigmoid(np.concatenate([x, h], axis=1) @ wg + bg)
    r, u = g[:, :hdim], g[:, hdim:]
    c = np.tanh(np.concatenate([x, r * h], axis=1) @ wc + bc)
    return (u * h + (1.0 - u) * c).astype(dtype)


def gru_cell_split(x, h, lay, dtype):
    """Same cell, split-weight form: rows [0,I) of W multiply x, rows [I,I+H) multiply h."""
    i_l = x.shape[1]
    hdim = h.shape[1]
    wg, wc = lay["Wg"].astype(dtype), lay["Wc"].astype(dtype)
    g = _sigmoid(x @ wg[:i_l] + h @ wg[i_l:] + lay["bg"].astype(dtype))
    r, u = g[:, :hdim], g[:, hdim:]
    c = np.tanh(x @ wc[:i_l] + (r * h) @ wc[i_l:] + lay["bc"].astype(dtype))
    return (u * h + (1.0 - u) * c).astype(dtype)


def _forward(cell, w, mel, state, seq_len, dtype, use_relu, value_clip):
    mel = np.asarray(mel, dtype=dtype)
    b, t_len, _ = mel.shape
    nl = len(w["layers"])
    hdim = w["Wfc"].shape[0]
    if state is None:
        state = np.zeros((nl, b, hdim), dtype)
    h = [np.array(state[l], dtype=dtype) for l in range(nl)]
    if seq_len is None:
        seq_len = np.full(b, t_len, np.int64)
    seq_len = np.asarray(seq_len)
    top = np.zeros((b, t_len, hdim), dtype)
    for t in range(t_len):
        live = (t < seq_len)[:, None]
        x = mel[:, t, :]
        new_h = []
        for l in range(nl):
            hn = cell(x, h[l], w["layers"][l], dtype)
            new_h.append(hn)
            x = hn
        # dynamic_rnn copy-through: finished rows keep every layer's state, emit zero output
        for l in range(nl):
            h[l] = np.where(live, new_h[l], h[l])
        top[:, t, :] = np.where(live, new_h[-1], 0.0)
    logits = top.reshape(-1, hdim) @ w["Wfc"].astype(dtype) + w["bfc"].astype(dtype)
    logits = logits.reshape(b, t_len, w["Wfc"].shape[1])
    if use_relu:                      # models/rnn_ctc.py:280-283
        logits = np.maximum(logits, 0.0)
        if value_clip > 0:
            logits = np.clip(logits, 0.0, 20.0)
    return logits.astype(dtype), np.stack(h).astype(dtype)


def gru_forward(w, mel, state=None, seq_len=None, dtype=np.float32, use_relu=False, value_clip=-1.0):
    """(mel[B,T,I], state[L,B,H]) -> (logits[B,T,C], state'[L,B,H]); models/rnn_ctc.py:155-163."""
    return _forward(gru_cell, w, mel, state, seq_len, dtype, use_relu, value_clip)


def gru_forward_split(w, mel, state=None, seq_len=None, dtype=np.float32):
    return _forward(gru_cell_split, w, mel, state, seq_len, dtype, False, -1.0)


def softmax(logits):
    """models/rnn_ctc.py:165 -- softmax over the class axis."""
    z = logits - logits.max(axis=-1, keepdims=True)
    e = np.exp(z)
    return (e / e.sum(axis=-1, keepdims=True)).astype(logits.dtype)


def stream_chunks(w, mel, chunk_lens, dtype=np.float32):
    """detector.py:190-196 / :280-285 -- feed consecutive chunks, round-trip the state."""
    b = mel.shape[0]
    nl, hdim = len(w["layers"]), w["Wfc"].shape[0]
    state = np.zeros((nl, b, hdim), dtype)
    outs, pos = [], 0
    for n in chunk_lens:
        lg, state = gru_forward(w, mel[:, pos:pos + n], state, dtype=dtype)
        outs.append(lg)
        pos += n
    return np.concatenate(outs, axis=1), state


# --------------------------------------------------------------------------------------
# bf16 variant (BASELINE configs[2]) -- restatement of WHERE the HIP bf16 path rounds, not a new
# reference: the reference has no bf16 model.  Matmul inputs (weights, x, h, r*h, top-layer h) are
# rounded to bf16 (nearest even); products are exact in fp32, accumulation / activations / state stay
# in higher precision.  PARITY UNPINNED like the fp32 GRU oracle.
# --------------------------------------------------------------------------------------
def bf16_round(a):
    a = np.ascontiguousarray(a, np.float32)
    u = a.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
    return r.view(np.float32).reshape(a.shape)


def f16_round(a):
    """fp32 -> fp16 (nearest even) -> fp32: ONE 11-bit piece of the f16x3 split (csrc/gru_f16x3.hip keeps two)."""
    return np.ascontiguousarray(a, np.float32).astype(np.float16).astype(np.float32)


def gru_forward_rounded(w, mel, round_w, round_x, state=None, seq_len=None):
    """The rounding model of gru_forward_bf16 with the two roundings chosen separately: round_w on every weight matrix,
    round_x on every matmul input (x, h, r*h, top-layer h).  Used to price operand precisions BEFORE a kernel is written
    (tests/test_oracle_gru.py::test_what_single_piece_fp16_operands_would_cost): with bf16_round for both it IS
    gru_forward_bf16, whose predictions the bf16 HIP stack reproduces (tests/test_gpu_bf16.py)."""
    mel = np.asarray(mel, np.float32)
    b, t_len, _ = mel.shape
    nl = len(w["layers"])
    hdim = w["Wfc"].shape[0]
    if state is None:
        state = np.zeros((nl, b, hdim), np.float32)
    h = [np.array(state[l], np.float64) for l in range(nl)]
    if seq_len is None:
        seq_len = np.full(b, t_len, np.int64)
    seq_len = np.asarray(seq_len)
    wq = [dict(Wg=round_w(l["Wg"]).astype(np.float64), Wc=round_w(l["Wc"]).astype(np.float64),
               bg=l["bg"].astype(np.float64), bc=l["bc"].astype(np.float64)) for l in w["layers"]]
    wfc = round_w(w["Wfc"]).astype(np.float64)
    top = np.zeros((b, t_len, hdim), np.float64)
    for t in range(t_len):
        live = (t < seq_len)[:, None]
        x = round_x(mel[:, t, :]).astype(np.float64)
        for l in range(nl):
            lay = wq[l]
            i_l = x.shape[1]
            hb = round_x(h[l].astype(np.float32)).astype(np.float64)
            g = _sigmoid(x @ lay["Wg"][:i_l] + hb @ lay["Wg"][i_l:] + lay["bg"])
            r, u = g[:, :hdim], g[:, hdim:]
            rh = round_x((r * h[l]).astype(np.float32)).astype(np.float64)
            c = np.tanh(x @ lay["Wc"][:i_l] + rh @ lay["Wc"][i_l:] + lay["bc"])
            hn = u * h[l] + (1.0 - u) * c
            h[l] = np.where(live, hn, h[l])
            x = round_x(h[l].astype(np.float32)).astype(np.float64)
            if l == nl - 1:
                top[:, t, :] = np.where(live, hn, 0.0)
    logits = round_x(top.reshape(-1, hdim).astype(np.float32)).astype(np.float64) @ wfc + w["bfc"].astype(np.float64)
    return logits.reshape(b, t_len, -1), np.stack(h)


def gru_forward_bf16(w, mel, state=None, seq_len=None):
    mel = np.asarray(mel, np.float32)
    b, t_len, _ = mel.shape
    nl = len(w["layers"])
    hdim = w["Wfc"].shape[0]
    if state is None:
        state = np.zeros((nl, b, hdim), np.float32)
    h = [np.array(state[l], np.float64) for l in range(nl)]
    if seq_len is None:
        seq_len = np.full(b, t_len, np.int64)
    seq_len = np.asarray(seq_len)
    wq = [dict(Wg=bf16_round(l["Wg"]).astype(np.float64), Wc=bf16_round(l["Wc"]).astype(np.float64),
               bg=l["bg"].astype(np.float64), bc=l["bc"].astype(np.float64)) for l in w["layers"]]
    wfc = bf16_round(w["Wfc"]).astype(np.float64)
    top = np.zeros((b, t_len, hdim), np.float64)
    for t in range(t_len):
        live = (t < seq_len)[:, None]
        x = bf16_round(mel[:, t, :]).astype(np.float64)
        for l in range(nl):
            lay = wq[l]
            i_l = x.shape[1]
            hb = bf16_round(h[l].astype(np.float32)).astype(np.float64)
            g = _sigmoid(x @ lay["Wg"][:i_l] + hb @ lay["Wg"][i_l:] + lay["bg"])
            r, u = g[:, :hdim], g[:, hdim:]
            rh = bf16_round((r * h[l]).astype(np.float32)).astype(np.float64)
            c = np.tanh(x @ lay["Wc"][:i_l] + rh @ lay["Wc"][i_l:] + lay["bc"])
            hn = u * h[l] + (1.0 - u) * c
            h[l] = np.where(live, hn, h[l])
            x = bf16_round(h[l].astype(np.float32)).astype(np.float64)       # next layer's input
            if l == nl - 1:
                top[:, t, :] = np.where(live, hn, 0.0)
    logits = bf16_round(top.reshape(-1, hdim).astype(np.float32)).astype(np.float64) @ wfc + w["bfc"].astype(np.float64)
    return logits.reshape(b, t_len, -1), np.stack(h)


# --------------------------------------------------------------------------------------
# int8 ("octbit") variant -- the graph octbit/octbit_graph.py produces from the deploy model:
# every MatMul whose node name passes default_octbit_matmul_name_check (:218-225) becomes an
# OctbitMatMul (octbit_mat_mul_op.cc) on weights quantised by octize_weight_int8_signed (:191-215).
# For the GRU that is the gates and candidate matmuls of cell_1.. (cell_0 is excluded by name) and
# the class projection `model/MatMul`; bias adds, sigmoid/tanh, the state update and the whole of
# layer 0 stay fp32.  The graph runs batch 1, so each GRU matmul is one op call on a [1, I+H] row
# (activation range = that row's min/max) and the projection is one call on the call's [T, H] block
# (one range over all T frames, zero rows of finished frames included).
# The op arithmetic is PINNED (octbit_ops_test.py known answers); the GRU wiring around it is
# PARITY UNPINNED like the rest of this file.
# --------------------------------------------------------------------------------------
def octbit_layer_is_quantised(layer):
    from oracle import octbit_oracle as Q
    return Q.default_octbit_matmul_name_check(
        "model/drnn/multi_rnn_cell/cell_%d/gru_cell/gates/MatMul" % layer)


def gru_forward_octbit(w, mel, state=None, seq_len=None, saturate=True):
    from oracle import octbit_oracle as Q
    f32 = np.float32
    mel = np.asarray(mel, f32)
    b, t_len, _ = mel.shape
    nl = len(w["layers"])
    hdim = w["Wfc"].shape[0]
    if state is None:
        state = np.zeros((nl, b, hdim), f32)
    h = [np.array(state[l], f32) for l in range(nl)]
    if seq_len is None:
        seq_len = np.full(b, t_len, np.int64)
    seq_len = np.asarray(seq_len)
    qw = []
    for l, lay in enumerate(w["layers"]):
        qw.append(dict(g=Q.octize_weight_int8_signed(lay["Wg"]), c=Q.octize_weight_int8_signed(lay["Wc"]))
                  if octbit_layer_is_quantised(l) else None)
    top = np.zeros((b, t_len, hdim), f32)
    for t in range(t_len):
        live = (t < seq_len)[:, None]
        x = mel[:, t, :]
        for l in range(nl):
            lay = w["layers"][l]
            if qw[l] is None:
                hn = gru_cell(x, h[l], lay, f32)
            else:
                (gq, gs, gb), (cq, cs, cb) = qw[l]["g"], qw[l]["c"]
                g = _sigmoid((Q.octbit_rows(np.concatenate([x, h[l]], 1), gq, gs, gb, saturate=saturate) + lay["bg"].astype(f32)).astype(f32))
                r, u = g[:, :hdim], g[:, hdim:]
                c = np.tanh((Q.octbit_rows(np.concatenate([x, (r * h[l]).astype(f32)], 1), cq, cs, cb, saturate=saturate)
                             + lay["bc"].astype(f32)).astype(f32))
                hn = (u * h[l] + (f32(1.0) - u) * c).astype(f32)
            h[l] = np.where(live, hn, h[l])
            x = hn
        top[:, t, :] = np.where(live, hn, f32(0))
    if Q.default_octbit_matmul_name_check("model/MatMul"):
        fq, fs, fb = Q.octize_weight_int8_signed(w["Wfc"])
        # pad the class rows are not needed: N is free, only K % 64 == 0 is required
        flat = Q.octbit_rows(top.reshape(b * t_len, hdim), fq, fs, fb, groups=np.repeat(np.arange(b), t_len), saturate=saturate)
    else:
        flat = top.reshape(-1, hdim) @ w["Wfc"].astype(f32)
    logits = (flat + w["bfc"].astype(f32)).astype(f32).reshape(b, t_len, -1)
    return logits, np.stack(h).astype(f32)
