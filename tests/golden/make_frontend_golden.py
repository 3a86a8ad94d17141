"""Generates tests/golden/frontend_golden.npz from a THIRD-PARTY implementation of the two pieces the deploy graph's
front-end is made of (models/rnn_ctc.py:134-149): a Slaney-scale, area-normalised mel filter bank
(librosa.filters.mel(sr=16000, n_fft=400, fmin=300, fmax=8000, n_mels), :139-141) and an un-windowed, un-padded
400/160 magnitude STFT (utils/stft.py:27-81 tf_frame + tf.spectral.rfft, :135-137) multiplied by it (:142-149).

librosa and TensorFlow cannot be installed here; `transformers.audio_utils` (installed in this image, 5.15) implements
the same published algorithms -- mel_filter_bank(norm="slaney", mel_scale="slaney") is librosa's htk=False/norm=1 bank
-- and was written by neither the reference's author nor this repo's; the un-windowed magnitude spectrum is taken a second
time from scipy.signal.stft (boxcar window, no boundary extension).  It is NOT the reference run here: the front-end
stays "parity partial"  until tests/golden/make_gru_golden.py case D runs under TF 1.x + librosa.

    python tests/golden/make_frontend_golden.py            (build container; writes the .npz next to this file)
"""
import os

import numpy as np
import scipy
import scipy.signal
from transformers import audio_utils as A
import transformers

HERE = os.path.dirname(os.path.abspath(__file__))
SR, NFFT, HOP, FMIN, FMAX = 16000, 400, 160, 300.0, 8000.0     # config/rnn_config.py:57-65


def bank(n_mels):
    """[201, n_mels] float64, the transpose of librosa.filters.mel's [n_mels, 201]."""
    return A.mel_filter_bank(num_frequency_bins=NFFT // 2 + 1, num_mel_filters=n_mels, min_frequency=FMIN,
                             max_frequency=FMAX, sampling_rate=SR, norm="slaney", mel_scale="slaney")


def melspec(pcm, fb):
    """[T, n_mels] float64: rectangular window, no centring/padding, |rfft|, power 1, mel matmul, no log."""
    s = A.spectrogram(np.asarray(pcm, np.float64), window=np.ones(NFFT), frame_length=NFFT, hop_length=HOP,
                      fft_length=NFFT, power=1.0, center=False, mel_filters=fb, mel_floor=0.0, dtype=np.float64)
    return s.T


def linspec_scipy(pcm):
    """[T, 201] float64 magnitude spectrum from a SECOND third-party implementation: scipy.signal.stft with a boxcar window,
    no boundary extension, no padding (it scales by 1 / sum(window) = 1 / 400, undone here)."""
    _, _, z = scipy.signal.stft(np.asarray(pcm, np.float64), fs=SR, window="boxcar", nperseg=NFFT, noverlap=NFFT - HOP, nfft=NFFT,
                                boundary=None, padded=False, return_onesided=True)
    return np.abs(z).T * NFFT


def main():
    rng = np.random.default_rng(20174)
    out = {"transformers_version": np.array(transformers.__version__), "scipy_version": np.array(scipy.__version__)}
    for n_mels in (40, 60):
        # the deploy graph holds the bank as a float32 constant (tf.constant of librosa's float array, :139-141)
        out["basis_%d" % n_mels] = bank(n_mels)
    t = np.arange(8000) / SR
    cases = {
        "noise_3600": rng.standard_normal(3600) * 0.085,                       # one detector.py chunk, speech-like level
        "noise_loud_3840": rng.standard_normal(3840) * 0.6,                    # the carry-extended chunk length
        "tone_8000": 0.3 * np.sin(2 * np.pi * 1234.5 * t) + 0.05 * np.sin(2 * np.pi * 5000.0 * t),
        "chirp_8000": 0.2 * np.sin(2 * np.pi * (300.0 * t + 3500.0 * t * t)),
        "int16_like_3600": np.round(rng.standard_normal(3600) * 3000.0) / 32768.0,
        "exact_400": rng.standard_normal(400) * 0.1,                           # exactly one frame
        "short_559": rng.standard_normal(559) * 0.1,                           # one frame, 159 samples left over
    }
    for name, pcm in cases.items():
        pcm = pcm.astype(np.float32)                                           # what the placeholder is fed
        out["pcm_" + name] = pcm
        out["lin_" + name] = linspec_scipy(pcm)
        for n_mels in (40, 60):
            fb32 = out["basis_%d" % n_mels].astype(np.float32).astype(np.float64)
            out["mel%d_%s" % (n_mels, name)] = melspec(pcm, fb32)
    np.savez_compressed(os.path.join(HERE, "frontend_golden.npz"), **out)
    for k, v in out.items():
        print(k, getattr(v, "shape", None))


if __name__ == "__main__":
    main()
