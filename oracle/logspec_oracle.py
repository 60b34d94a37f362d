"""CPU restatement of the reference's audio front end — TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows ``datasets/preprocessing.py:158-186`` (``LogSpectrogram.__call__``):

    spect = |librosa.stft(sig, n_fft=2*n_fft, hop_length=int(hop_size*sr))|**2            (:175)
    spect = concat(spect[:1], mean over bin pairs of spect[1:])                           (:176)
    spect = spect[:, :int(duration*rate)]                                                 (:177-179)
    spect = librosa.power_to_db(spect, top_db=100)                                        (:181)
    spect = (spect - mean[:, None]) / (std[:, None] + 1e-5)           if normalize        (:182-183)
    return spect.T[None]                                               -> [1, T, n_fft/2 + 1]   (:184-186)

**Parity unpinned.**  The arithmetic lives in librosa, which is neither vendored in /root/reference nor pinned
in its conda-spec-list.txt, and is not installed in this image; the reference holds no test vectors for this
path.  This file restates librosa's published algorithm as of the 0.7/0.8 releases contemporary with the
reference (2020): ``stft`` = centred frames (signal reflect-padded by n_fft/2 on both sides), periodic Hann
window of length n_fft (scipy.signal.get_window('hann', n_fft, fftbins=True)), rfft per frame, frames at
multiples of the hop, ``1 + len(sig) // hop`` of them; ``power_to_db(S, ref=1.0, amin=1e-10, top_db)`` =
``10*log10(max(amin, S)) - 10*log10(max(amin, ref))`` floored at ``max - top_db``.  Everything in float64
except where the reference's dtype flow says float32 (the input signal).
"""
import numpy as np


def stft_power(sig, n_fft, hop):
    """|STFT|^2, librosa.stft(center=True, pad_mode='reflect', window='hann') -> [n_fft//2 + 1, 1 + len//hop]."""
    sig = np.asarray(sig, dtype=np.float64)
    pad = n_fft // 2
    x = np.pad(sig, (pad, pad), mode="reflect")
    n_frames = 1 + (len(x) - n_fft) // hop
    k = np.arange(n_fft)
    win = 0.5 - 0.5 * np.cos(2.0 * np.pi * k / n_fft)               # periodic Hann
    idx = np.arange(n_frames)[:, None] * hop + k[None, :]
    frames = x[idx] * win[None, :]
    spec = np.fft.rfft(frames, axis=1)                               # [T, n_fft//2 + 1]
    return (spec.real ** 2 + spec.imag ** 2).T


def power_to_db(S, top_db=100.0, amin=1e-10, ref=1.0):
    log_spec = 10.0 * np.log10(np.maximum(amin, S)) - 10.0 * np.log10(np.maximum(amin, ref))
    if top_db is not None:
        log_spec = np.maximum(log_spec, log_spec.max() - top_db)
    return log_spec


def log_spectrogram(sig, sr, n_fft=512, hop_size=0.005, duration=None, mean=None, std=None):
    """sig: [1, nsamples] (mono, as AudioPrep returns it, preprocessing.py:150).  Returns ([1, T, n_fft/2+1], rate)."""
    rate = 1.0 / hop_size
    hop = int(hop_size * sr)
    spect = stft_power(np.asarray(sig)[0], 2 * n_fft, hop)
    spect = np.concatenate([spect[:1], spect[1:].reshape(n_fft // 2, 2, -1).mean(1)], 0)
    if duration is not None:
        spect = spect[:, :int(duration * rate)]
    spect = power_to_db(spect, top_db=100.0)
    if mean is not None:
        spect = (spect - np.asarray(mean, dtype=np.float64)[:, None]) / (np.asarray(std, dtype=np.float64)[:, None] + 1e-5)
    return spect.T[None].astype(np.float32), rate
