"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the deploy graph's audio front-end.

PARITY PARTIAL (third-party fixture only: tests/golden/frontend_golden.npz from transformers.audio_utils -- the same Slaney mel
bank and un-windowed magnitude STFT, tests/test_frontend_golden.py; not the reference run here).
models/rnn_ctc.py:134-149 builds it from tf.spectral.rfft and librosa.filters.mel; neither
TensorFlow nor librosa (era 0.5, not pinned by the reference) is installable here and the reference holds no
fixture for this stage.  Restated from the call sites:
  utils/stft.py:27-81        tf_frame: num_frames = 1 + floor((N - 400)/160), no padding, no window
  models/rnn_ctc.py:137      |rfft(frames, 400)|
  models/rnn_ctc.py:139-149  matmul with librosa.filters.mel(sr=16000, n_fft=400, fmin=300, fmax=8000, n_mels).T
and librosa 0.5's published algorithm for filters.mel (htk=False, norm=1: Slaney scale, area normalisation).
Only tests/ may import this module."""
import numpy as np


def hz_to_mel(f):
    f = np.asarray(f, np.float64)
    f_sp, min_log_hz = 200.0 / 3.0, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, f / f_sp)


def mel_to_hz(m):
    m = np.asarray(m, np.float64)
    f_sp, min_log_hz = 200.0 / 3.0, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_basis(sr=16000, n_fft=400, n_mels=40, fmin=300.0, fmax=8000.0):
    nf = 1 + n_fft // 2
    fftfreqs = np.linspace(0.0, sr / 2.0, nf)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    w = np.zeros((n_mels, nf))
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        w[i] = np.maximum(0.0, np.minimum(lower, upper))
    w *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
    return w


def frames(pcm, n_fft=400, hop=160):
    pcm = np.asarray(pcm)
    n = pcm.shape[-1]
    t = 0 if n < n_fft else 1 + (n - n_fft) // hop
    idx = np.arange(n_fft)[None, :] + hop * np.arange(t)[:, None]
    return pcm[..., idx]


def melspec(pcm, sr=16000, n_fft=400, hop=160, n_mels=40, fmin=300.0, fmax=8000.0):
    """pcm [B,N] -> [B,T,n_mels] (float64 arithmetic)."""
    fr = frames(np.asarray(pcm, np.float64), n_fft, hop)
    lin = np.abs(np.fft.rfft(fr, n_fft, axis=-1))
    return lin @ mel_basis(sr, n_fft, n_mels, fmin, fmax).astype(np.float32).astype(np.float64).T
