"""DeployModel -- the streaming inference graph of models/rnn_ctc.py:113-166 as one object backed
by the HIP kernels.

    model = DeployModel(config, weights)
    logits, next_state = model.step(mel_chunk, prev_state)        # the north-star surface

or, in the reference's own vocabulary (detector.py:190-193), with the feed the shipped graph takes --
`model/inputX:0` is the 1-D PCM chunk (models/rnn_ctc.py:130-134) and the front-end
(tf_frame -> |rfft| -> mel matmul, :134-149) runs inside the graph:

    softmax, state = model.run(['model/softmax:0', 'model/rnn_states:0'],
                               {'model/inputX:0': data, 'model/rnn_initial_states:0': state})

The commented mel-input variant of the graph (models/rnn_ctc.py:150-153: inputX = [T, n_mel]) is accepted on
the same name: a 2-D feed is one stream's mel, a 3-D feed is B streams' mel.

Tensors are torch CUDA tensors and stay on the device; B independent streams are batched:
mel [B,T,n_mel], state [L,B,H], logits/softmax [B,T,C].
"""
import ctypes

import numpy as np
import torch

from . import _lib
from . import weights as _weights

FEED_INPUT = "model/inputX:0"
FEED_STATE = "model/rnn_initial_states:0"
FETCH_SOFTMAX = "model/softmax:0"
FETCH_LOGIT = "model/logit:0"
FETCH_STATE = "model/rnn_states:0"
FETCH_TOKENS = "model/ctc_decode2_tokens:0"     # extension: fused per-frame ctc_decode2 events


class DeployModel(object):
    def __init__(self, config, weights, device="cuda:0", kernel="auto"):
        self.config = config
        self._frontend = None
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.InvalidArgumentError(-1, "DeployModel needs a CUDA/HIP device, got %s" % device)
        self._lib = _lib.load()
        blob = weights if isinstance(weights, np.ndarray) else _weights.to_blob(config, weights)
        blob = np.ascontiguousarray(blob, np.float32)
        self._cfg = _lib.KwsConfig(config.n_mel, config.hidden_size, config.num_layers, config.num_classes,
                                   int(bool(config.use_relu)), float(config.value_clip),
                                   {"fp32": _lib.FP32, "bf16": _lib.BF16, "int8": _lib.INT8, "f16x3": _lib.F16X3}[getattr(config, "precision", "fp32")])
        self._handle = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self._lib.kws_create(ctypes.byref(self._cfg), blob.ctypes.data_as(ctypes.c_void_p),
                                            blob.nbytes, ctypes.byref(self._handle)))
        self.set_kernel(kernel)

    # -- lifecycle ---------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_frontend", None) is not None:
            self._frontend.close()
            self._frontend = None
        if getattr(self, "_handle", None) is not None and self._handle.value:
            self._lib.kws_destroy(self._handle)
            self._handle = ctypes.c_void_p()

    @property
    def frontend(self):
        """The in-graph audio front-end (models/rnn_ctc.py:134-149), created on first use."""
        if self._frontend is None:
            from .frontend import MelFrontend
            self._frontend = MelFrontend(self.config, device=self.device)
        return self._frontend

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_kernel(self, kind):
        k = {"auto": _lib.KERNEL_AUTO, "generic": _lib.KERNEL_GENERIC, "resident": _lib.KERNEL_RESIDENT}[kind]
        _lib.check(self._lib.kws_set_kernel(self._handle, k))
        self.kernel = kind

    def reserve(self, batch, frames):
        _lib.check(self._lib.kws_reserve(self._handle, int(batch), int(frames)))

    def scratch_stats(self):
        """(bytes reserved for inter-layer seams, device (re)allocations so far -- each one synchronised)."""
        nbytes, allocs = ctypes.c_size_t(), ctypes.c_int32()
        _lib.check(self._lib.kws_scratch_stats(self._handle, ctypes.byref(nbytes), ctypes.byref(allocs)))
        return int(nbytes.value), int(allocs.value)

    def status(self):
        """Raises if a finished asynchronous step of this handle failed on the device (kws_poll_error)."""
        _lib.check(self._lib.kws_poll_error(self._handle))

    def set_profiling(self, enable):
        _lib.check(self._lib.kws_set_profiling(self._handle, int(bool(enable))))

    def kernel_times(self, reset=True):
        """[(ms_sum, launches)] per layer kernel since the last reset (synchronises)."""
        n = self.config.num_layers
        ms = (ctypes.c_float * n)()
        cnt = (ctypes.c_int32 * n)()
        _lib.check(self._lib.kws_kernel_times(self._handle, ms, cnt, int(reset)))
        return [(float(ms[i]), int(cnt[i])) for i in range(n)]

    def kernel_names(self):
        """Kernel(s) the last forward() launched, per profiling slot ('' = ran inside another slot's launch)."""
        out = []
        for l in range(self.config.num_layers):
            buf = ctypes.create_string_buffer(160)
            _lib.check(self._lib.kws_last_launch(self._handle, l, buf, len(buf)))
            out.append(buf.value.decode())
        return out

    def selftest(self):
        """kws_selftest: the kernels this model launches against the library's own known answers (TensorFlow's published
        GRUCell constants + a host double-precision loop); raises on mismatch.  KWS_SELFTEST=1 runs it in every create."""
        with torch.cuda.device(self.device):
            _lib.check(self._lib.kws_selftest(self._handle))

    # -- state -------------------------------------------------------------------------------
    def zero_state(self, batch=1):
        """detector.py:123-124 / clean_state :313-316, for `batch` streams."""
        return torch.zeros(self.config.num_layers, batch, self.config.hidden_size,
                           dtype=torch.float32, device=self.device)

    def fresh_prev_word(self, batch=1):
        return torch.full((batch,), -1, dtype=torch.int32, device=self.device)

    # -- compute -----------------------------------------------------------------------------
    def forward(self, mel, state, seq_len=None, reset_mask=None, want_logits=True, want_softmax=True,
                prev_word=None, decode2_thres=0.4, state_out=None, out=None):
        """One sess.run of the deploy graph on B streams.  Returns a dict with the requested
        'logits', 'softmax', 'state' and, if prev_word is given, 'tokens' (prev_word is updated
        in place)."""
        cfg = self.config
        mel = self._dev(mel, torch.float32, "mel")
        state = self._dev(state, torch.float32, "state")
        if mel.dim() != 3 or mel.shape[2] != cfg.n_mel:
            raise _lib.InvalidArgumentError(-1, "mel must be [B,T,%d], got %s" % (cfg.n_mel, tuple(mel.shape)))
        b, t = int(mel.shape[0]), int(mel.shape[1])
        if tuple(state.shape) != (cfg.num_layers, b, cfg.hidden_size):
            raise _lib.InvalidArgumentError(-1, "state must be [%d,%d,%d], got %s"
                                            % (cfg.num_layers, b, cfg.hidden_size, tuple(state.shape)))
        if seq_len is not None:
            seq_len = self._dev(seq_len, torch.int32, "seq_len")
            if tuple(seq_len.shape) != (b,):
                raise _lib.InvalidArgumentError(-1, "seq_len must be [%d]" % b)
        if reset_mask is not None:
            reset_mask = self._dev(reset_mask, torch.uint8, "reset_mask")
            if tuple(reset_mask.shape) != (b,):
                raise _lib.InvalidArgumentError(-1, "reset_mask must be [%d]" % b)
        out = out or {}
        c = cfg.num_classes
        logits = out.get("logits") if want_logits else None
        if want_logits and logits is None:
            logits = torch.empty(b, t, c, dtype=torch.float32, device=self.device)
        softmax = out.get("softmax") if want_softmax else None
        if want_softmax and softmax is None:
            softmax = torch.empty(b, t, c, dtype=torch.float32, device=self.device)
        tokens = None
        if prev_word is not None:
            prev_word = self._dev(prev_word, torch.int32, "prev_word")
            tokens = out.get("tokens")
            if tokens is None:
                tokens = torch.empty(b, t, dtype=torch.int8, device=self.device)
        if state_out is None:
            state_out = torch.empty_like(state)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.kws_step(
                self._handle, _lib.ptr(mel), _lib.ptr(state), _lib.ptr(logits), _lib.ptr(softmax),
                _lib.ptr(state_out), _lib.ptr(seq_len), _lib.ptr(reset_mask), _lib.ptr(tokens),
                _lib.ptr(prev_word), float(decode2_thres), b, t, _lib.current_stream_ptr()))
        res = {"state": state_out}
        if want_logits:
            res["logits"] = logits
        if want_softmax:
            res["softmax"] = softmax
        if tokens is not None:
            res["tokens"] = tokens
        return res

    def step(self, mel_chunk, prev_state, **kw):
        """(mel_chunk, prev_state) -> (logits, next_state)."""
        r = self.forward(mel_chunk, prev_state, want_logits=True, want_softmax=False, **kw)
        return r["logits"], r["state"]

    def run(self, fetches, feed_dict):
        """tf.Session.run on the frozen graph's tensor names (main.py:339-342).
        model/inputX:0 is, as in the shipped graph, the 1-D float PCM chunk (models/rnn_ctc.py:130-134;
        detector.py:190-193 feeds `data`): framed, transformed and projected on the mel basis on the device
        (kws_frontend_run) before kws_step.  Fewer than fft_size samples give zero frames (tf_frame,
        utils/stft.py:27-81) and the state comes back unchanged.  A 2-D feed [T,n_mel] is the commented
        mel-input variant (:150-153) at batch 1, a 3-D feed [B,T,n_mel] is B streams (state [L,B,H])."""
        single = isinstance(fetches, str)
        names = [fetches] if single else list(fetches)
        known = (FETCH_SOFTMAX, FETCH_LOGIT, FETCH_STATE)
        for n in names:
            if n not in known:
                raise _lib.InvalidArgumentError(-1, "unknown fetch %r (graph exports %s)" % (n, ", ".join(known)))
        for k in feed_dict:
            if k not in (FEED_INPUT, FEED_STATE):
                raise _lib.InvalidArgumentError(-1, "unknown feed %r" % k)
        if FEED_INPUT not in feed_dict or FEED_STATE not in feed_dict:
            raise _lib.InvalidArgumentError(-1, "feeds %s and %s are required" % (FEED_INPUT, FEED_STATE))
        mel = torch.as_tensor(feed_dict[FEED_INPUT])
        if mel.dim() == 1:
            if not mel.dtype.is_floating_point:
                raise _lib.InvalidArgumentError(-1, "model/inputX:0 is a float32 placeholder, got %s "
                                                "(convert PCM with detector.buf_to_float)" % mel.dtype)
            mel = self.frontend.forward(mel)                # [T, n_mel]
        if mel.dim() not in (2, 3):
            raise _lib.InvalidArgumentError(-1, "model/inputX:0 must be PCM [N], mel [T,%d] or mel [B,T,%d]; got rank %d"
                                            % (self.config.n_mel, self.config.n_mel, mel.dim()))
        squeeze = mel.dim() == 2
        if squeeze:
            mel = mel.unsqueeze(0)
        r = self.forward(mel, feed_dict[FEED_STATE], want_logits=FETCH_LOGIT in names,
                         want_softmax=FETCH_SOFTMAX in names)
        # SURVEY 8a footnote 1: consumers index model/softmax:0 as [T,C] (detector.py:197,285) and
        # model/logit:0 as [1,T,C] (detector.py:249-250) at batch 1
        table = {FETCH_STATE: r["state"], FETCH_LOGIT: r.get("logits"),
                 FETCH_SOFTMAX: (r["softmax"][0] if squeeze else r["softmax"]) if "softmax" in r else None}
        outs = [table[n] for n in names]
        return outs[0] if single else outs

    def _dev(self, x, dtype, name):
        t = torch.as_tensor(x)
        if t.dtype != dtype:
            if dtype == torch.float32 and t.dtype in (torch.float64, torch.float16, torch.bfloat16):
                t = t.to(dtype)
            elif dtype in (torch.int32, torch.uint8) and not t.dtype.is_floating_point:
                t = t.to(dtype)
            else:
                raise _lib.InvalidArgumentError(-1, "%s must be %s, got %s" % (name, dtype, t.dtype))
        if t.device != self.device:
            t = t.to(self.device)
        return t.contiguous()
