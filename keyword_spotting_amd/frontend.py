"""MelFrontend -- the in-graph audio front-end of models/rnn_ctc.py:134-149 on the GPU (kws_frontend_*):
tf_frame(400, 160) -> |rfft(., 400)| -> matmul with librosa.filters.mel(...)^T, power 1, no window."""
import ctypes

import numpy as np
import torch

from . import _lib


class MelFrontend(object):
    def __init__(self, config, device="cuda:0"):
        self.config = config
        self.device = torch.device(device)
        self._lib = _lib.load()
        self._cfg = _lib.KwsFrontendConfig(int(config.samplerate), int(config.fft_size), int(config.hop_size),
                                           int(config.n_mel), float(config.fmin), float(config.fmax))
        self._handle = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self._lib.kws_frontend_create(ctypes.byref(self._cfg), ctypes.byref(self._handle)))

    def close(self):
        if getattr(self, "_handle", None) is not None and self._handle.value:
            self._lib.kws_frontend_destroy(self._handle)
            self._handle = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def num_frames(self, n_samples):
        return int(self._lib.kws_frontend_frames(ctypes.byref(self._cfg), int(n_samples)))

    def mel_basis(self):
        """[n_mel, fft/2+1] fp32 -- librosa.filters.mel layout (the graph uses its transpose)."""
        out = np.empty((self.config.n_mel, self.config.fft_size // 2 + 1), np.float32)
        _lib.check(self._lib.kws_frontend_mel_basis(self._handle, out.ctypes.data_as(ctypes.c_void_p)))
        return out

    def forward(self, pcm):
        """pcm [B,N] (or [N]) float -> mel [B,T,n_mel] (or [T,n_mel]) on the device."""
        x = torch.as_tensor(pcm)
        if x.dtype != torch.float32:
            x = x.to(torch.float32)
        single = x.dim() == 1
        if single:
            x = x.unsqueeze(0)
        if x.dim() != 2:
            raise _lib.InvalidArgumentError(-1, "expected signal to have rank 2 but was %d" % x.dim())
        x = x.to(self.device).contiguous()
        b, n = int(x.shape[0]), int(x.shape[1])
        t = self.num_frames(n)
        mel = torch.empty(b, t, self.config.n_mel, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.kws_frontend_run(self._handle, _lib.ptr(x), b, n, _lib.ptr(mel),
                                                  _lib.current_stream_ptr()))
        return mel[0] if single else mel

    def forward_carry(self, carry, chunk, n_next):
        """Streaming form (detector.py:179-183): mel of [carry | chunk] without building the concatenation, and the
        next carry (its last n_next samples).  carry [B,n_c] or None, chunk [B,n] float32 device tensors."""
        chunk = chunk.contiguous()
        b, n = int(chunk.shape[0]), int(chunk.shape[1])
        nc = 0 if carry is None else int(carry.shape[1])
        if carry is not None:
            carry = carry.contiguous()
        t = self.num_frames(nc + n)
        mel = torch.empty(b, t, self.config.n_mel, dtype=torch.float32, device=self.device)
        nxt = torch.empty(b, int(n_next), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.kws_frontend_run_carry(self._handle, _lib.ptr(carry) if nc else None, nc, _lib.ptr(chunk), n, b,
                                                        _lib.ptr(mel), _lib.ptr(nxt), int(n_next), _lib.current_stream_ptr()))
        return mel, nxt
