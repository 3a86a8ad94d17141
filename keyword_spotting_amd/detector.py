"""HotwordDetector -- the streaming state-carry loop of detector.py:104-316 without the audio I/O
(pyaudio capture, wav dumps and matplotlib plots are host I/O with no arithmetic and are out of scope).

What is reproduced, per sess.run-sized chunk and per stream:
  * zero-initialised recurrent state owned by the caller of the model (detector.py:123-124)
  * VAD gate: a silent chunk resets the state and clears the decode window BEFORE the chunk is run
    (detector.py:168-177; vad(data, 30))
  * model run with the carried state, state replaced by the returned one (detector.py:190-196)
  * 15-chunk sliding window of softmax chunks (SimpleQueue(15), detector.py:122,195,197)
  * ctc_decode2 over the whole concatenated window, ctc_predict(result, '1233') (detector.py:200-201)
  * on trigger: callback, window cleared, state reset (detector.py:202-209)
  * sample carry arithmetic between PCM chunks (detector.py:179-183) -- ChunkFramer
  * test2: chunked replay of a whole utterance, one ctc_decode at the end (detector.py:254-289)

B independent streams run in lock step (one kws_step per chunk for all of them); the window
bookkeeping is per stream.  feed() takes the mel variant of the graph (models/rnn_ctc.py:150-153); feed_pcm() is
the shipped graph's contract (PCM in, front-end on the device).
"""
import numpy as np
import torch

from . import _lib
from .basic_vad import vad as _vad
from .prediction import decode_batch
from .prediction import ctc_predict as _ctc_predict
from .queue import SimpleQueue


def buf_to_float(x, n_bytes=2, dtype=torch.float32):
    """detector.py:40-43 (RingBuffer.get, :74-79): little-endian signed PCM -> float in [-1, 1), scale 2^-(8n-1).
    Integer tensors stay on their device, so 16-bit PCM crosses PCIe at half the bytes and is widened there
    (exact: a power-of-two scale)."""
    x = torch.as_tensor(x)
    if x.dtype.is_floating_point:
        return x.to(dtype)
    want = {2: torch.int16, 4: torch.int32, 1: torch.int8}[n_bytes]
    if x.dtype != want:
        raise _lib.InvalidArgumentError(-1, "expected %s PCM for n_bytes=%d, got %s" % (want, n_bytes, x.dtype))
    return x.to(dtype) * (1.0 / float(1 << (8 * n_bytes - 1)))


class ChunkFramer(object):
    """Sample bookkeeping of detector.py:179-183: how many frames a PCM chunk yields once the carried
    tail of the previous chunk is prepended, and how many samples are carried forward."""

    def __init__(self, fft_size=400, hop_size=160):
        self.fft_size, self.hop_size = fft_size, hop_size
        self.carry = 0

    def push(self, n_samples):
        total = self.carry + n_samples
        frames = 0 if total < self.fft_size else (total - self.fft_size) // self.hop_size + 1
        self.carry = (total - self.fft_size) % self.hop_size + (self.fft_size - self.hop_size)
        return frames


class HotwordDetector(object):
    def __init__(self, model, batch=1, window_chunks=15, vad_thres=30, label=None, decode_thres=0.4,
                 detected_callback=None):
        self.model = model
        self.config = model.config
        self.batch = int(batch)
        self.vad_thres = vad_thres
        self.label = label or self.config.label_seqs
        self.decode_thres = decode_thres
        self.detected_callback = detected_callback
        self.prob_queue = [SimpleQueue(window_chunks) for _ in range(self.batch)]
        self.state = model.zero_state(self.batch)
        self.reset_next = torch.zeros(self.batch, dtype=torch.uint8, device=model.device)
        self.triggers = [0] * self.batch

    # detector.py:313-316 -- on the device the reset is a per-stream mask consumed by the next kws_step
    def clean_state(self, which=None):
        if which is None:
            self.reset_next.fill_(1)
        else:
            self.reset_next[which] = 1

    def feed(self, mel_chunk, pcm_chunk=None, speech=None):
        """One loop iteration of detector.py:158-209 for every stream.
        mel_chunk [B,T,n_mel] (or [T,n_mel] when batch == 1).  The VAD decision comes from `speech`
        ([B] bool) or is computed from `pcm_chunk` ([B,N]) with vad(data, vad_thres); neither -> speech.
        Returns the list of streams that triggered on this chunk."""
        mel = torch.as_tensor(mel_chunk)
        if mel.dim() == 2:
            mel = mel.unsqueeze(0)
        if mel.shape[0] != self.batch:
            raise _lib.InvalidArgumentError(-1, "expected %d streams, got %d" % (self.batch, mel.shape[0]))
        if speech is None and pcm_chunk is not None:
            pcm = torch.as_tensor(pcm_chunk)
            speech = _vad(pcm if pcm.dim() == 2 else pcm.unsqueeze(0), self.vad_thres).bool().cpu().numpy()
        if speech is not None:
            for b in np.nonzero(~np.asarray(speech, bool))[0]:       # :171-177
                self.clean_state(int(b))
                self.prob_queue[int(b)].clear()
        # (zero frames: dynamic_rnn hands the state back, clean_state() has zeroed it where the mask says so)
        r = self.model.forward(mel, self.state, reset_mask=self.reset_next, want_logits=False, want_softmax=True,
                               state_out=self.state)
        self.reset_next.zero_()
        softmax = r["softmax"]
        for b in range(self.batch):                                  # :195
            self.prob_queue[b].add(softmax[b])
        windows = [torch.cat(q.get_all(), 0) for q in self.prob_queue]   # :197
        lens = torch.tensor([w.shape[0] for w in windows], dtype=torch.int32)
        tmax = int(lens.max()) if self.batch else 0
        padded = torch.zeros(self.batch, max(tmax, 1), self.config.num_classes, device=self.model.device)
        for b, w in enumerate(windows):
            padded[b, :w.shape[0]] = w
        words, counts = decode_batch(_lib.DECODE2, padded, lens, 3, self.decode_thres, 0.0)   # :200
        hits = _ctc_predict((words, counts), self.label).cpu().numpy()                        # :201
        fired = []
        for b in np.nonzero(hits)[0]:
            b = int(b)
            fired.append(b)
            self.triggers[b] += 1
            if self.detected_callback is not None:
                self.detected_callback(b)
            self.prob_queue[b].clear()                               # :203
            self.clean_state(b)                                      # :208
        return fired

    def feed_pcm(self, pcm_chunk, frontend):
        """The reference's real input contract (detector.py:162-193): raw PCM chunks [B,n].  Prepends the
        carried tail (:179), keeps the new tail (:181-183), runs the in-graph front-end
        (models/rnn_ctc.py:134-149, here `frontend` = keyword_spotting_amd.frontend.MelFrontend) and the
        loop body; VAD looks at the newly captured chunk only (:168)."""
        chunk = torch.as_tensor(pcm_chunk)
        if chunk.dim() == 1:
            chunk = chunk.unsqueeze(0)
        chunk = buf_to_float(chunk.to(self.model.device))             # int16 PCM -> [-1, 1) as RingBuffer.get does (:74-79)
        if not hasattr(self, "res"):
            self.res = chunk[:, :0]                                   # :125
        if chunk.shape[1] == 0:                                       # :164-166: an empty read is skipped
            return []
        data = torch.cat([self.res, chunk], 1)                        # :179
        fft, hop = self.config.fft_size, self.config.hop_size
        n = int(data.shape[1])
        if n < fft:
            # not a full frame yet: the reference still runs the whole iteration -- vad / clean_state / queue.clear (:168-177),
            # every sample carried (:181-183 keeps them all), sess.run over zero frames, an empty softmax into the queue (:195)
            self.res = data
            mel = torch.zeros(self.batch, 0, self.config.n_mel, device=self.model.device)
            return self.feed(mel, pcm_chunk=chunk)
        keep = (n - fft) % hop + (fft - hop)                          # :181-182
        self.res = data[:, n - keep:].contiguous()                    # :183
        mel = frontend.forward(data.contiguous())
        return self.feed(mel, pcm_chunk=chunk)

    def test(self, pcm, frontend, label=None):
        """detector.py:214-229 without the file and plot I/O: whole utterances [B,N] (or [N]) of PCM in ONE run from the
        detector's current state -- front-end, GRU stack -- then ctc_decode over the whole softmax and ctc_predict.
        Returns (hit [B] int32, (words, counts), softmax [B,T,C], logits [B,T,C]); the state is left untouched, as there."""
        x = torch.as_tensor(pcm)
        if x.dim() == 1:
            x = x.unsqueeze(0)
        x = buf_to_float(x.to(self.model.device))
        mel = frontend.forward(x.contiguous())
        state = self.state if x.shape[0] == self.batch else self.model.zero_state(int(x.shape[0]))
        r = self.model.forward(mel, state, want_logits=True, want_softmax=True)
        decoded = decode_batch(_lib.DECODE, r["softmax"], None, 3, 0.5, 0.2)                     # :224
        return _ctc_predict(decoded, label or self.label), decoded, r["softmax"], r["logits"]    # :226

    def test2(self, mel, chunk_frames):
        """detector.py:254-289: replay whole utterances [B,T,n_mel] in chunks with the state threaded
        through, accumulate the softmax, decode once with ctc_decode.  Returns (words, counts)."""
        mel = torch.as_tensor(mel)
        if mel.dim() == 2:
            mel = mel.unsqueeze(0)
        mel = mel.to(self.model.device)
        state = self.model.zero_state(mel.shape[0])                  # :263
        parts, pos = [], 0
        for n in chunk_frames:
            r = self.model.forward(mel[:, pos:pos + n].contiguous(), state, want_logits=False, want_softmax=True)
            state = r["state"]                                       # :284
            parts.append(r["softmax"])                               # :285
            pos += n
        accu = torch.cat(parts, 1)
        return decode_batch(_lib.DECODE, accu, None, 3, 0.5, 0.2)    # :288


class StreamManager(object):
    """The same loop with every per-stream decision on the device.  feed_pcm is ONE native call
    per chunk (kws_stream_feed: VAD gate -> front-end with sample carry -> GRU stack -> 15-chunk window with windowed
    ctc_decode2 + ctc_predict, trigger -> clear + restart; the window step rides inside the last GRU layer's launch); feed
    takes mel chunks and chains kws_step and kws_window_step_incremental itself.  No per-stream host work; results are identical to HotwordDetector
    (tests/test_gpu_detector.py, tests/test_gpu_frontend.py)."""

    def __init__(self, model, batch, window_chunks=15, max_frames=32, vad_thres=30, label=None, decode_thres=0.4):
        import ctypes
        self.model, self.config, self.batch = model, model.config, int(batch)
        self.vad_thres, self.decode_thres = vad_thres, decode_thres
        self.label = (label or self.config.label_seqs).encode()
        self._lib = _lib.load()
        self._win = ctypes.c_void_p()
        with torch.cuda.device(model.device):
            _lib.check(self._lib.kws_window_create(self.batch, int(window_chunks), int(max_frames),
                                                   self.config.num_classes, float(decode_thres), ctypes.byref(self._win)))
        dev = model.device
        self.state = model.zero_state(self.batch)
        self.restart = torch.zeros(self.batch, dtype=torch.uint8, device=dev)     # reset requested by a trigger
        self.hit = torch.zeros(self.batch, dtype=torch.int32, device=dev)
        self.max_frames = int(max_frames)
        self._stream, self._stream_frontend, self._stream_fe_handle = None, None, None

    def _close_stream(self):
        if getattr(self, "_stream", None) is not None and self._stream.value:
            self._lib.kws_stream_destroy(self._stream)
        self._stream, self._stream_frontend, self._stream_fe_handle = None, None, None

    def close(self):
        self._close_stream()
        if getattr(self, "_win", None) is not None and self._win.value:
            self._lib.kws_window_destroy(self._win)
            self._win.value = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def feed(self, mel_chunk, pcm_chunk=None, speech=None):
        """-> hit [B] int32 device tensor (1 = keyword detected on this chunk)."""
        mel = torch.as_tensor(mel_chunk)
        if mel.dim() == 2:
            mel = mel.unsqueeze(0)
        dev = self.model.device
        if speech is None and pcm_chunk is not None:
            pcm = torch.as_tensor(pcm_chunk)
            speech = _vad(pcm if pcm.dim() == 2 else pcm.unsqueeze(0), self.vad_thres)
        if speech is None:
            silent = torch.zeros(self.batch, dtype=torch.uint8, device=dev)
        else:
            silent = (torch.as_tensor(speech).to(dev) == 0).to(torch.uint8)
        reset = torch.maximum(self.restart, silent)                       # detector.py:171-177 and :208
        r = self.model.forward(mel, self.state, reset_mask=reset, want_logits=False, want_softmax=True,
                               state_out=self.state)
        sm = r["softmax"]
        with torch.cuda.device(dev):
            # the incremental form of the window step: the state kws_stream_feed (feed_pcm) keeps, so mel-fed and PCM-fed
            # chunks may alternate on one manager
            _lib.check(self._lib.kws_window_step_incremental(self._win, _lib.ptr(sm), int(sm.shape[1]), _lib.ptr(silent), self.label,
                                                             _lib.ptr(self.hit), _lib.ptr(self.restart), _lib.current_stream_ptr()))
        return self.hit

    def feed_pcm(self, pcm_chunk, frontend):
        """pcm_chunk [B, n]: float samples, or int16 PCM as the sound card delivers it (widened on the device as
        buf_to_float does, detector.py:74-79).  One native call per chunk (kws_stream_feed): VAD + reset masks, the
        front-end on [carried samples | chunk] (detector.py:179-183, never concatenated), the GRU stack on the carried
        state, the window / decode / trigger step.  -> hit [B] int32 device tensor."""
        import ctypes
        chunk = torch.as_tensor(pcm_chunk)
        if chunk.dim() == 1:
            chunk = chunk.unsqueeze(0)
        if chunk.shape[0] != self.batch:
            raise _lib.InvalidArgumentError(-1, "expected %d streams, got %d" % (self.batch, chunk.shape[0]))
        if chunk.dtype == torch.int16:
            is_i16 = 1
        elif chunk.dtype.is_floating_point:
            is_i16, chunk = 0, chunk.to(torch.float32)
        else:
            raise _lib.InvalidArgumentError(-1, "expected float or int16 PCM, got %s" % chunk.dtype)
        chunk = chunk.to(self.model.device).contiguous()
        dev = self.model.device
        if not self.model._handle.value or not frontend._handle.value:
            raise _lib.InvalidArgumentError(-1, "the model or the front-end has been closed")
        if self._stream is None or self._stream_frontend is not frontend or self._stream_fe_handle != frontend._handle.value:
            self._close_stream()
            self._stream = ctypes.c_void_p()
            with torch.cuda.device(dev):
                _lib.check(self._lib.kws_stream_create(self.model._handle, frontend._handle, self._win, self.batch,
                                                       self.max_frames * int(self.config.hop_size), float(self.vad_thres),
                                                       self.label, _lib.ptr(self.state), _lib.ptr(self.restart),
                                                       ctypes.byref(self._stream)))
            self._stream_frontend, self._stream_fe_handle = frontend, frontend._handle.value
        with torch.cuda.device(dev):
            _lib.check(self._lib.kws_stream_feed(self._stream, _lib.ptr(chunk), int(chunk.shape[1]), is_i16, _lib.ptr(self.hit),
                                                 _lib.current_stream_ptr()))
        return self.hit
