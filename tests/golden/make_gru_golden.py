"""Generates tests/golden/gru_tf_golden.npz by RUNNING THE REFERENCE'S OWN GRAPH CODE -- models/rnn_ctc.py:202-284
(inference1: GRUCell x L under MultiRNNCell + dynamic_rnn; inference2: the class projection) and :165 (softmax) --
under TensorFlow 1.x.  This is the pin the GRU / dense / softmax oracle is missing: the
arithmetic lives in tensorflow.contrib.rnn.GRUCell, a third-party dependency that is neither vendored in the reference
nor installable in the build image, so until a machine with a TF-1.x wheel runs

    python tests/golden/make_gru_golden.py [--reference /root/reference] [--out tests/golden/gru_tf_golden.npz]

the fixture does not exist, tests/test_tf_golden.py skips with "PARITY UNPINNED", and the oracle is checked only
against independent restatements.  Once the .npz is committed, the same tests compare BOTH the oracle (CPU) and the HIP
path (GPU) with TensorFlow's own outputs.  The fixture holds data only: the variables TF initialised, the inputs fed
and the outputs fetched.

Requirements on the generating machine: tensorflow 1.1-1.15 (tf.contrib.rnn.GRUCell; the reference's era is 1.1-1.3)
and librosa (models/rnn_ctc.py:26 imports it at module level).  Nothing is stubbed: if either import fails the script
says so and exits with status 2.

What is generated (every case: seeded variables via the reference's own initialisers, seeded inputs):
  A     config A  (n_mel=40, H=128, L=2, C=6): batch 1, 300 frames, zero state            (BASELINE configs[0])
  A5    config A: batch 5, 64 frames, ragged sequence_length incl. 0 and T, random initial state
  Achk  config A: batch 1, chunks of 21/22/23 frames with the state threaded through (detector.py:190-196)
  B     config B  (n_mel=60): batch 3, 50 frames
  C     config C  (n_mel=60, H=256, L=4): batch 2, 40 frames                               (BASELINE configs[4])
  Arelu config A with use_relu + value_clip (models/rnn_ctc.py:280-283): batch 2, 30 frames
  D     the shipped DeployModel graph itself (models/rnn_ctc.py:113-166), PCM in: 4 chunks of 3600 samples with the
        sample carry of detector.py:179-183 -- pins the front-end (tf_frame, rfft, librosa mel basis) as well
"""
import argparse
import os
import sys

EXIT_UNAVAILABLE = 2


def _import_reference(ref_root):
    sys.dont_write_bytecode = True
    os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
    if not os.path.isdir(ref_root):
        print("make_gru_golden: reference tree %s not found" % ref_root, file=sys.stderr)
        raise SystemExit(EXIT_UNAVAILABLE)
    try:
        import tensorflow as tf
    except Exception as exc:                           # noqa: BLE001
        print("make_gru_golden: tensorflow is not importable here (%s) -- GRU parity stays UNPINNED" % exc, file=sys.stderr)
        raise SystemExit(EXIT_UNAVAILABLE)
    if not hasattr(tf, "contrib"):
        print("make_gru_golden: tensorflow %s has no tf.contrib (need 1.x: the reference uses tf.contrib.rnn.GRUCell)"
              % tf.__version__, file=sys.stderr)
        raise SystemExit(EXIT_UNAVAILABLE)
    sys.path.insert(0, ref_root)
    try:
        from models import rnn_ctc                    # the reference module itself
    except Exception as exc:                           # noqa: BLE001  (librosa missing, API drift ...)
        print("make_gru_golden: cannot import the reference's models/rnn_ctc.py (%s)" % exc, file=sys.stderr)
        raise SystemExit(EXIT_UNAVAILABLE)
    return tf, rnn_ctc


class _Cfg(object):
    """The attributes inference1/inference2/get_cell/DeployModel read (config/rnn_config.py:57-99)."""

    def __init__(self, n_mel, hidden, layers, use_relu=False, value_clip=-1):
        self.hidden_size, self.num_layers, self.num_classes = hidden, layers, 6
        self.freq_size = self.n_mel = n_mel
        self.use_layer_norm = self.use_residual = False
        self.keep_prob, self.variational_recurrent = 1.0, False
        self.use_relu, self.value_clip = use_relu, value_clip
        self.batch_size = 1
        self.samplerate, self.fft_size, self.hop_size, self.fmin, self.fmax = 16000, 400, 160, 300, 8000


def _variables(tf, sess, scope):
    out = {}
    for v in tf.global_variables():
        if v.name.startswith(scope + "/"):
            out[v.name[len(scope) + 1:]] = sess.run(v)
    return out


def _stack_case(tf, rnn_ctc, np, tag, cfg, batch, frames, seed, seq_len=None, state0=None, chunks=None):
    """inference1 + inference2 + softmax on a mel placeholder (the graph DeployModel builds after its front-end,
    models/rnn_ctc.py:155-165, for `batch` streams)."""
    rng = np.random.default_rng(seed)
    mel = (np.abs(rng.standard_normal((batch, frames, cfg.n_mel))) * 2).astype(np.float32)
    if state0 is None:
        state0 = np.zeros((cfg.num_layers, batch, cfg.hidden_size), np.float32)
    if seq_len is None:
        seq_len = np.full(batch, frames, np.int32)
    g = tf.Graph()
    with g.as_default():
        tf.set_random_seed(seed)
        with tf.variable_scope("model"):
            x = tf.placeholder(tf.float32, [batch, None, cfg.n_mel], name="mel")
            s0 = tf.placeholder(tf.float32, [cfg.num_layers, batch, cfg.hidden_size], name="rnn_initial_states")
            sl = tf.placeholder(tf.int32, [batch], name="seq")
            outputs, states = rnn_ctc.inference1(cfg, x, sl, is_training=False, initial_state=tuple(tf.unstack(s0)))
            states = tf.stack(states)
            logits = rnn_ctc.inference2(outputs, cfg, batch)
            softmax = tf.nn.softmax(logits)
        with tf.Session(graph=g, config=tf.ConfigProto(device_count={"GPU": 0})) as sess:
            sess.run(tf.global_variables_initializer())
            # GRUCell's default kernels at this fan-in are small; the fixtures should exercise the non-linearities
            variables = _variables(tf, sess, "model")
            res = {}
            if chunks is None:
                lg, sm, st = sess.run([logits, softmax, states], {x: mel, s0: state0, sl: seq_len})
            else:
                st, lgs, sms, pos = state0, [], [], 0
                for n in chunks:
                    lg_, sm_, st = sess.run([logits, softmax, states],
                                            {x: mel[:, pos:pos + n], s0: st, sl: np.full(batch, n, np.int32)})
                    lgs.append(lg_); sms.append(sm_); pos += n
                lg, sm = np.concatenate(lgs, 1), np.concatenate(sms, 1)
                res[tag + "/chunks"] = np.asarray(chunks, np.int32)
    res.update({tag + "/mel": mel, tag + "/state0": state0, tag + "/seq_len": seq_len.astype(np.int32),
                tag + "/logits": lg, tag + "/softmax": sm, tag + "/state": st,
                tag + "/shape": np.asarray([cfg.n_mel, cfg.hidden_size, cfg.num_layers, cfg.num_classes,
                                            int(cfg.use_relu), int(cfg.value_clip)], np.int32)})
    for name, arr in variables.items():
        res[tag + "/var/" + name.replace(":0", "")] = arr
    return res


def _deploy_case(tf, rnn_ctc, np, tag, cfg, seed):
    """The shipped graph: DeployModel (models/rnn_ctc.py:113-166) fed PCM chunk by chunk as detector.py:179-196 does."""
    rng = np.random.default_rng(seed)
    pcm = (rng.standard_normal(3600 * 4) * 0.3).astype(np.float32)
    g = tf.Graph()
    res = {}
    with g.as_default():
        tf.set_random_seed(seed)
        with tf.variable_scope("model"):
            rnn_ctc.DeployModel(cfg)
        with tf.Session(graph=g, config=tf.ConfigProto(device_count={"GPU": 0})) as sess:
            sess.run(tf.global_variables_initializer())
            variables = _variables(tf, sess, "model")
            state = np.zeros((cfg.num_layers, 1, cfg.hidden_size), np.float32)
            carry = pcm[:0]
            for c in range(4):
                data = np.concatenate((carry, pcm[3600 * c:3600 * (c + 1)]), 0)
                keep = (len(data) - cfg.fft_size) % cfg.hop_size + (cfg.fft_size - cfg.hop_size)
                carry = data[-keep:]
                sm, lg, state = sess.run(["model/softmax:0", "model/logit:0", "model/rnn_states:0"],
                                         {"model/inputX:0": data, "model/rnn_initial_states:0": state})
                res["%s/chunk%d/data" % (tag, c)] = data
                res["%s/chunk%d/softmax" % (tag, c)] = sm
                res["%s/chunk%d/logit" % (tag, c)] = lg
                res["%s/chunk%d/state" % (tag, c)] = state
            try:
                res[tag + "/mel_basis"] = sess.run(g.get_tensor_by_name("model/Const:0"))
            except Exception:                          # noqa: BLE001  -- name differs across TF versions; optional
                pass
    res[tag + "/shape"] = np.asarray([cfg.n_mel, cfg.hidden_size, cfg.num_layers, cfg.num_classes, 0, -1], np.int32)
    for name, arr in variables.items():
        res[tag + "/var/" + name.replace(":0", "")] = arr
    return res


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(here, "gru_tf_golden.npz"))
    args = ap.parse_args()
    tf, rnn_ctc = _import_reference(args.reference)
    import numpy as np
    out = {"meta/tensorflow_version": np.asarray(tf.__version__), "meta/generator": np.asarray("tests/golden/make_gru_golden.py")}
    rng = np.random.default_rng(7001)
    out.update(_stack_case(tf, rnn_ctc, np, "A", _Cfg(40, 128, 2), 1, 300, 7002))
    seq = np.asarray([0, 64, 17, 40, 63], np.int32)
    st5 = (0.5 * rng.standard_normal((2, 5, 128))).astype(np.float32)
    out.update(_stack_case(tf, rnn_ctc, np, "A5", _Cfg(40, 128, 2), 5, 64, 7003, seq_len=seq, state0=st5))
    out.update(_stack_case(tf, rnn_ctc, np, "Achk", _Cfg(40, 128, 2), 1, 66, 7004, chunks=[21, 22, 23]))
    out.update(_stack_case(tf, rnn_ctc, np, "B", _Cfg(60, 128, 2), 3, 50, 7005))
    out.update(_stack_case(tf, rnn_ctc, np, "C", _Cfg(60, 256, 4), 2, 40, 7006))
    out.update(_stack_case(tf, rnn_ctc, np, "Arelu", _Cfg(40, 128, 2, use_relu=True, value_clip=1), 2, 30, 7007))
    try:
        out.update(_deploy_case(tf, rnn_ctc, np, "D", _Cfg(40, 128, 2), 7008))
    except Exception as exc:                           # noqa: BLE001 -- the stack cases above are the pin; say what is missing
        print("make_gru_golden: DeployModel case skipped (%s)" % exc, file=sys.stderr)
    np.savez_compressed(args.out, **out)
    print("wrote %s (%d arrays, tensorflow %s)" % (args.out, len(out), tf.__version__))


if __name__ == "__main__":
    main()
