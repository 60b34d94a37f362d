"""Batched log-spectrogram on the GPU — drop-in for ``datasets.preprocessing.LogSpectrogram``
(reference ``datasets/preprocessing.py:158-186``).

Same constructor and ``__call__(sig, sr, duration=None) -> (spect, rate)`` contract; additionally accepts a
batch ``[B, 1, nsamples]`` (or ``[B, nsamples]``) of mono clips already on the GPU and then returns
``[B, 1, T, n_fft/2 + 1]`` — the layout ``main-avid.py`` feeds the audio encoder — in one call
(``avid_logspec``: frames -> DFT as one fp32-MFMA GEMM -> power / pair-mean / dB / top_db floor / z-score).
The reference runs librosa per clip on 36-72 CPU workers (Cross-N1024.yaml:3).
"""
import numpy as np
import torch

from avid_hip import ops

_STATS = {(512, 24000): "datasets/assets/audio-spectDB-24k-513-norm-stats.npz",
          (256, 24000): "datasets/assets/audio-spectDB-24k-257-norm-stats.npz"}


class LogSpectrogram(object):
    def __init__(self, fps, n_fft=512, hop_size=0.005, normalize=False, device="cuda", stats=None):
        self.inp_fps = fps
        self.n_fft = n_fft
        self.hop_size = hop_size
        self.rate = 1. / hop_size
        self.normalize = normalize
        self.device = torch.device(device)
        self.mean = self.std = None
        if self.normalize:
            # the reference reads these files relative to its checkout (preprocessing.py:167-171); ``stats`` may
            # name another .npz with 'mean' / 'std' arrays of n_fft/2 + 1 bins
            path = stats if stats is not None else _STATS[(n_fft, int(fps))]
            st = np.load(path)
            self.mean = torch.as_tensor(st['mean'], dtype=torch.float32, device=self.device).contiguous()
            self.std = torch.as_tensor(st['std'], dtype=torch.float32, device=self.device).contiguous()

    def __call__(self, sig, sr, duration=None):
        single = False
        if isinstance(sig, np.ndarray):
            sig = torch.from_numpy(np.ascontiguousarray(sig, dtype=np.float32))
        if sig.dim() == 2 and sig.shape[0] == 1:           # the reference's [1, nsamples] (AudioPrep output)
            single = True
        elif sig.dim() == 3:                               # [B, 1, nsamples]
            sig = sig[:, 0]
        sig = sig.to(self.device, torch.float32).contiguous()
        hop_length = int(self.hop_size * sr)
        frames = 1 + sig.shape[1] // hop_length
        if duration is not None:
            frames = min(frames, int(duration * self.rate))
        spect = ops.log_spectrogram(sig, 2 * self.n_fft, hop_length, frames, self.mean, self.std, top_db=100.)
        return (spect[0] if single else spect), self.rate
