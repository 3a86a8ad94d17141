"""Weight containers for the streaming GRU: the canonical (TF-variable-layout) dict and the flat
fp32 blob the C ABI takes.

dict layout (one entry per tf.get_variable of the reference graph):
  layers[l] = {Wg [I_l+H, 2H]  drnn/.../cell_l/gru_cell/gates/kernel      (gate order r, u)
               bg [2H]         .../gates/bias        (TF initialises to 1.0)
               Wc [I_l+H, H]   .../candidate/kernel
               bc [H]          .../candidate/bias}
  Wfc [H, C] weightsClasses, bfc [C] biasesClasses   (models/rnn_ctc.py:265-273)
"""
import numpy as np


def layer_in_dims(config):
    return [config.n_mel if l == 0 else config.hidden_size for l in range(config.num_layers)]


def init_weights(config, seed=0):
    """Random-init weights of the reference architecture (no checkpoints exist offline):
    glorot-uniform kernels (TF default for GRUCell), gate bias 1, candidate bias 0, fc truncated
    normal (models/rnn_ctc.py:266), fc bias 0."""
    rng = np.random.default_rng(seed)
    h, c = config.hidden_size, config.num_classes
    layers = []
    for i_l in layer_in_dims(config):
        k = i_l + h
        a = np.sqrt(6.0 / (k + 2 * h))
        wg = rng.uniform(-a, a, size=(k, 2 * h)).astype(np.float32)
        a = np.sqrt(6.0 / (k + h))
        wc = rng.uniform(-a, a, size=(k, h)).astype(np.float32)
        layers.append(dict(Wg=wg, bg=np.ones(2 * h, np.float32), Wc=wc, bc=np.zeros(h, np.float32)))
    wfc = rng.standard_normal((h, c))
    bad = np.abs(wfc) > 2.0
    while bad.any():
        wfc[bad] = rng.standard_normal(int(bad.sum()))
        bad = np.abs(wfc) > 2.0
    return dict(layers=layers, Wfc=wfc.astype(np.float32), bfc=np.zeros(c, np.float32))


def check_shapes(config, w):
    h, c = config.hidden_size, config.num_classes
    if len(w["layers"]) != config.num_layers:
        raise ValueError("expected %d layers, got %d" % (config.num_layers, len(w["layers"])))
    for l, (lay, i_l) in enumerate(zip(w["layers"], layer_in_dims(config))):
        want = dict(Wg=(i_l + h, 2 * h), bg=(2 * h,), Wc=(i_l + h, h), bc=(h,))
        for name, shape in want.items():
            if tuple(lay[name].shape) != shape:
                raise ValueError("layer %d %s has shape %s, expected %s" % (l, name, lay[name].shape, shape))
    if tuple(w["Wfc"].shape) != (h, c) or tuple(w["bfc"].shape) != (c,):
        raise ValueError("fc weights have shapes %s %s, expected %s %s"
                         % (w["Wfc"].shape, w["bfc"].shape, (h, c), (c,)))


def to_blob(config, w):
    check_shapes(config, w)
    parts = []
    for lay in w["layers"]:
        parts += [lay["Wg"], lay["bg"], lay["Wc"], lay["bc"]]
    parts += [w["Wfc"], w["bfc"]]
    return np.ascontiguousarray(np.concatenate([np.asarray(p, np.float32).ravel() for p in parts]))


def from_blob(config, blob):
    blob = np.asarray(blob, np.float32).ravel()
    h, c = config.hidden_size, config.num_classes
    pos, layers = 0, []

    def take(*shape):
        nonlocal pos
        n = int(np.prod(shape))
        out = blob[pos:pos + n].reshape(shape).copy()
        pos += n
        return out

    for i_l in layer_in_dims(config):
        layers.append(dict(Wg=take(i_l + h, 2 * h), bg=take(2 * h), Wc=take(i_l + h, h), bc=take(h)))
    w = dict(layers=layers, Wfc=take(h, c), bfc=take(c))
    if pos != blob.size:
        raise ValueError("blob has %d floats, config needs %d" % (blob.size, pos))
    return w


def save_npz(path, w):
    flat = {"Wfc": w["Wfc"], "bfc": w["bfc"]}
    for l, lay in enumerate(w["layers"]):
        for k, v in lay.items():
            flat["l%d_%s" % (l, k)] = v
    np.savez(path, **flat)


def load_npz(path):
    z = np.load(path)
    n = 1 + max(int(k[1:].split("_")[0]) for k in z.files if k.startswith("l"))
    return dict(layers=[{k: z["l%d_%s" % (l, k)] for k in ("Wg", "bg", "Wc", "bc")} for l in range(n)],
                Wfc=z["Wfc"], bfc=z["bfc"])


def from_tf_variables(config, variables):
    """Canonical dict from a {TF variable name: array} mapping (e.g. a checkpoint dumped to .npz).

    Names as the reference graph creates them (models/rnn_ctc.py:236-243 scope "drnn", :265-273):
      [model/]drnn/multi_rnn_cell/cell_<l>/gru_cell/{gates,candidate}/{kernel,bias}     TF >= 1.2
      ...                                            {gates,candidate}/{weights,biases}  TF 1.0-1.1
      [model/]weightsClasses, [model/]biasesClasses
    Optimiser slots (".../Adam", ".../Adam_1") and a trailing ":0" are ignored."""
    import re
    pat = re.compile(r"(?:^|/)cell_(\d+)/gru_cell/(gates|candidate)/(kernel|weights|bias|biases)(?::0)?$")
    slot = {("gates", "kernel"): "Wg", ("gates", "weights"): "Wg", ("gates", "bias"): "bg", ("gates", "biases"): "bg",
            ("candidate", "kernel"): "Wc", ("candidate", "weights"): "Wc", ("candidate", "bias"): "bc",
            ("candidate", "biases"): "bc"}
    layers = [dict() for _ in range(config.num_layers)]
    w = dict(layers=layers)
    for name, arr in variables.items():
        m = pat.search(name)
        if m:
            l = int(m.group(1))
            if l >= config.num_layers:
                raise ValueError("variable %s belongs to layer %d, config has %d layers" % (name, l, config.num_layers))
            key = slot[(m.group(2), m.group(3))]
            if key in layers[l]:
                raise ValueError("two variables map to layer %d %s (second: %s)" % (l, key, name))
            layers[l][key] = np.asarray(arr, np.float32)
            continue
        base = name[:-2] if name.endswith(":0") else name
        base = base.rsplit("/", 1)[-1]
        if base == "weightsClasses":
            w["Wfc"] = np.asarray(arr, np.float32)
        elif base == "biasesClasses":
            w["bfc"] = np.asarray(arr, np.float32)
    for l, lay in enumerate(layers):
        missing = [k for k in ("Wg", "bg", "Wc", "bc") if k not in lay]
        if missing:
            raise ValueError("layer %d: no variable found for %s" % (l, ", ".join(missing)))
    for k in ("Wfc", "bfc"):
        if k not in w:
            raise ValueError("no variable found for %s (weightsClasses / biasesClasses)" % k)
    check_shapes(config, w)
    return w


def to_tf_variables(w, new_names=True, prefix="model/"):
    """Inverse of from_tf_variables (for round-trip tests and for exporting back)."""
    kn, bn = ("kernel", "bias") if new_names else ("weights", "biases")
    out = {prefix + "weightsClasses": w["Wfc"], prefix + "biasesClasses": w["bfc"]}
    for l, lay in enumerate(w["layers"]):
        base = "%sdrnn/multi_rnn_cell/cell_%d/gru_cell/" % (prefix, l)
        out[base + "gates/" + kn], out[base + "gates/" + bn] = lay["Wg"], lay["bg"]
        out[base + "candidate/" + kn], out[base + "candidate/" + bn] = lay["Wc"], lay["bc"]
    return out
