"""SURVEY §8(f) rank 4 — the log-spectrogram front end.

CPU: the oracle's STFT restatement against scipy.signal.stft (an independent implementation; librosa itself is
neither vendored nor pinned by the reference, so this row stays "parity unpinned" — oracle/logspec_oracle.py).
GPU: avid_logspec (through datasets.gpu_audio.LogSpectrogram, the reference class's drop-in) against the oracle
on seeded signals.  Tolerance (fp32 DFT-by-GEMM over 1024 terms against float64; the reference's own output is
float32 dB): 5e-3 dB on every bin within 60 dB of the clip's maximum, 0.1 dB on the weaker ones, where the fp32
accumulation error of the strong partials shows (absolute power error <= 1e-6 of the clip maximum)."""
import numpy as np
import pytest
import torch

from oracle import logspec_oracle as L


def _signal(seed, n, kind):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / 24000.0
    if kind == "noise":
        x = 0.1 * rng.standard_normal(n)
    elif kind == "chirp":
        x = 0.3 * np.sin(2 * np.pi * (200.0 + 2500.0 * t) * t) + 0.01 * rng.standard_normal(n)
    else:   # speech-like: decaying harmonics + noise
        x = sum(0.2 / k * np.sin(2 * np.pi * 110.0 * k * t + k) for k in range(1, 20)) * np.exp(-1.5 * t)
        x = x + 0.003 * rng.standard_normal(n)
    return x.astype(np.float32)[None]


def _close_db(got, ref, strong_from=None):
    base = ref if strong_from is None else strong_from
    strong = base >= base.max() - 60.0
    err = np.abs(got - ref)
    assert err[strong].max() < 5e-3, float(err[strong].max())
    assert err.max() < 0.1, float(err.max())


def test_oracle_stft_matches_scipy():
    from scipy.signal import stft
    sig = _signal(0, 48000, "chirp")
    for n, hop in ((1024, 240), (512, 120), (512, 77)):
        x = np.pad(sig[0].astype(np.float64), (n // 2, n // 2), mode="reflect")
        _, _, Z = stft(x, window="hann", nperseg=n, noverlap=n - hop, nfft=n, boundary=None, padded=False)
        win = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(n) / n)
        ref = (np.abs(Z) * win.sum()) ** 2
        got = L.stft_power(sig[0], n, hop)
        assert got.shape == ref.shape == (n // 2 + 1, 1 + 48000 // hop)
        assert np.abs(got - ref).max() <= 1e-12 * ref.max()


def test_oracle_shapes_and_db_floor():
    sig = _signal(1, 48000, "speech")
    out, rate = L.log_spectrogram(sig, 24000, n_fft=512, hop_size=0.01, duration=2.0)
    assert out.shape == (1, 200, 257) and rate == 100.0 and out.dtype == np.float32
    assert out.max() - out.min() <= 100.0 + 1e-4           # top_db = 100 floor (preprocessing.py:181)
    # a pure tone lands in the expected (pair-averaged) bin: 3000 Hz -> STFT bin 128 -> output bin 64
    t = np.arange(48000) / 24000.0
    tone = np.sin(2 * np.pi * 3000.0 * t).astype(np.float32)[None]
    o2, _ = L.log_spectrogram(tone, 24000, n_fft=512, hop_size=0.01, duration=2.0)
    assert int(o2[0, 100].argmax()) == 64


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["noise", "chirp", "speech"])
@pytest.mark.parametrize("cfg", [(512, 0.01, 48000, 2.0), (256, 0.005, 24000, None), (512, 0.01, 31337, 1.0)])
def test_logspec_vs_oracle(kind, cfg, gpu_device):
    from datasets.gpu_audio import LogSpectrogram
    n_fft, hop_size, nsamp, duration = cfg
    rng = np.random.default_rng(5)
    F = n_fft // 2 + 1
    mean = rng.uniform(-30, -10, F).astype(np.float32)
    std = rng.uniform(5, 15, F).astype(np.float32)
    sigs = [_signal(10 + i, nsamp, kind) for i in range(3)]
    plain = LogSpectrogram(24000, n_fft=n_fft, hop_size=hop_size, normalize=False, device=gpu_device)
    # single clip, the reference's calling convention
    got, rate = plain(sigs[0], 24000, duration)
    ref, rrate = L.log_spectrogram(sigs[0], 24000, n_fft, hop_size, duration)
    assert rate == rrate and tuple(got.shape) == ref.shape
    _close_db(got.cpu().numpy(), ref)
    # batch, normalised
    norm = LogSpectrogram(24000, n_fft=n_fft, hop_size=hop_size, normalize=False, device=gpu_device)
    norm.mean = torch.from_numpy(mean).to(gpu_device)
    norm.std = torch.from_numpy(std).to(gpu_device)
    batch = torch.from_numpy(np.stack(sigs)).to(gpu_device)         # [B, 1, L]
    gotb, _ = norm(batch, 24000, duration)
    for i, s in enumerate(sigs):
        refn, _ = L.log_spectrogram(s, 24000, n_fft, hop_size, duration, mean=mean, std=std)
        assert gotb[i].shape == refn.shape
        raw, _ = L.log_spectrogram(s, 24000, n_fft, hop_size, duration)
        _close_db(gotb[i].cpu().numpy() * (std + 1e-5) + mean, refn * (std + 1e-5) + mean, strong_from=raw)
