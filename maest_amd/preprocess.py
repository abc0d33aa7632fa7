"""AugmentMelSTFT: the second log-mel parameterisation of the reference (models/preprocess.py:17-128), as one
instance of the fused HIP mel front end (csrc/mel2.hip).  north_star names the file; the reference's MAEST path
never calls it, and its arithmetic lives in torch.stft + torchaudio.compliance.kaldi
(torchaudio is absent here): PARITY UNPINNED -- the oracle restates the published algorithms.

Same constructor and forward contract as the reference class: waveform [B, S] (32 kHz) -> [B, n_mels, T],
T = 1 + (S - 1) // hopsize; in training mode the mel band edges are jittered (same two ``torch.randint`` calls
in the same order) and frequency / time stripes are zeroed before the affine normalisation.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn as nn

from . import ops


def kaldi_mel_banks(num_bins: int, n_fft: int, sr: float, low_freq: float, high_freq: float) -> np.ndarray:
    """``torchaudio.compliance.kaldi.get_mel_banks`` without VTLN warping (warp factor 1.0): triangular bands that
    are linear in the mel domain mel(f) = 1127 ln(1 + f / 700).  Returns float32 [num_bins, n_fft // 2]."""
    num_fft_bins = n_fft // 2
    nyquist = 0.5 * sr
    if high_freq <= 0.0:
        high_freq += nyquist
    assert 0.0 <= low_freq < nyquist and 0.0 < high_freq <= nyquist and low_freq < high_freq
    mel = lambda f: 1127.0 * np.log(1.0 + f / 700.0)
    fft_bin_width = sr / n_fft
    mel_low, mel_high = mel(low_freq), mel(high_freq)
    delta = (mel_high - mel_low) / (num_bins + 1)
    b = np.arange(num_bins, dtype=np.float32)[:, None]
    left = np.float32(mel_low) + b * np.float32(delta)
    center = np.float32(mel_low) + (b + 1.0) * np.float32(delta)
    right = np.float32(mel_low) + (b + 2.0) * np.float32(delta)
    m = mel(np.float32(fft_bin_width) * np.arange(num_fft_bins, dtype=np.float32))[None, :].astype(np.float32)
    up = (m - left) / (center - left)
    down = (right - m) / (right - center)
    return np.maximum(0.0, np.minimum(up, down)).astype(np.float32)


def sparse_bands(fb: np.ndarray):
    """[bands, bins] -> (start int32 [bands], len int32 [bands], weights fp32 [bands, stride], stride)."""
    starts, lens = [], []
    for row in fb:
        nz = np.nonzero(row)[0]
        starts.append(int(nz[0]) if len(nz) else 0)
        lens.append(int(nz[-1] - nz[0] + 1) if len(nz) else 0)
    stride = max(8, int(math.ceil(max(lens) / 8) * 8))
    w = np.zeros((fb.shape[0], stride), np.float32)
    for m, row in enumerate(fb):
        w[m, : lens[m]] = row[starts[m]: starts[m] + lens[m]]
    return np.asarray(starts, np.int32), np.asarray(lens, np.int32), w, stride


class AugmentMelSTFT(nn.Module):
    def __init__(self, n_mels=128, sr=32000, win_length=800, hopsize=320, n_fft=1024, freqm=48, timem=192, htk=False,
                 fmin=0.0, fmax=None, norm=1, fmin_aug_range=1, fmax_aug_range=1000):
        super().__init__()
        if n_fft != 1024 or hopsize != 320 or n_mels > 128:
            raise NotImplementedError("csrc/mel2.hip is instantiated for n_fft=1024, hopsize=320, n_mels<=128")
        self.win_length, self.n_mels, self.n_fft, self.sr, self.htk, self.fmin = win_length, n_mels, n_fft, sr, htk, fmin
        if fmax is None:
            fmax = sr // 2 - fmax_aug_range // 2
        self.fmax, self.norm, self.hopsize = fmax, norm, hopsize
        assert fmin_aug_range >= 1 and fmax_aug_range >= 1
        self.fmin_aug_range, self.fmax_aug_range = fmin_aug_range, fmax_aug_range
        self.freqm, self.timem = freqm, timem
        n = np.arange(win_length, dtype=np.float64)
        hann = 0.5 - 0.5 * np.cos(2.0 * np.pi * n / (win_length - 1))          # torch.hann_window(periodic=False)
        window = np.zeros(n_fft, np.float32)
        left = (n_fft - win_length) // 2                                           # torch.stft centres the window
        window[left:left + win_length] = hann.astype(np.float32)
        k = np.arange(n_fft, dtype=np.float64)
        tw = np.stack([np.cos(-2.0 * np.pi * k / n_fft), np.sin(-2.0 * np.pi * k / n_fft)], 1).astype(np.float32)
        self.register_buffer("window", torch.from_numpy(window), persistent=False)
        self.register_buffer("twiddle", torch.from_numpy(tw).contiguous(), persistent=False)
        self.register_buffer("preemphasis_coefficient", torch.as_tensor([[[-0.97, 1.0]]]), persistent=False)
        self._fb_cache = {}

    def _bands(self, fmin, fmax, device):
        key = (float(fmin), float(fmax), str(device))
        if key not in self._fb_cache:
            if len(self._fb_cache) > 64:
                self._fb_cache.clear()
            fb = kaldi_mel_banks(self.n_mels, self.n_fft, self.sr, fmin, fmax)     # [n_mels, 512]; bin 512 weighs 0
            st, ln, w, stride = sparse_bands(fb)
            self._fb_cache[key] = (torch.from_numpy(st).to(device), torch.from_numpy(ln).to(device),
                                   torch.from_numpy(w).to(device), stride)
        return self._fb_cache[key]

    def _stripes(self, B, size, param, n=1):
        """torchaudio ``mask_along_axis`` on a 3-D batch: ONE stripe shared by the batch; draws rand(1) twice."""
        value = torch.rand(1) * param
        min_value = torch.rand(1) * (size - value)
        start, end = int(min_value.long()), int(min_value.long()) + int(value.long())
        return torch.tensor([[[start, end - start]]] * B, dtype=torch.int32)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        assert isinstance(x, torch.Tensor) and x.dim() == 2, "AugmentMelSTFT expects a waveform batch [B, S]"
        fmin = self.fmin + torch.randint(self.fmin_aug_range, (1,)).item()
        fmax = self.fmax + self.fmax_aug_range // 2 - torch.randint(self.fmax_aug_range, (1,)).item()
        if not self.training:                          # don't augment eval data
            fmin, fmax = self.fmin, self.fmax
        st, ln, w, stride = self._bands(fmin, fmax, x.device)
        c = self.preemphasis_coefficient.reshape(-1).tolist()
        xw = x.float().contiguous()
        win, tw = self.window.to(x.device), self.twiddle.to(x.device)
        if not self.training:                          # log, then (melspec + 4.5) / 5 inside the kernel
            return ops.augment_mel(xw, win, tw, st, ln, w, stride, self.n_mels, c[0], c[1], 0.00001, 4.5, 5.0)
        mel = ops.augment_mel(xw, win, tw, st, ln, w, stride, self.n_mels, c[0], c[1], 0.00001, 0.0, 1.0)
        B, F, T = mel.shape
        fs = self._stripes(B, F, self.freqm) if self.freqm else None      # freqm first, then timem (:126-127)
        ts = self._stripes(B, T, self.timem) if self.timem else None
        ops.spec_mask_(mel, None if ts is None else ts.to(mel.device), None if fs is None else fs.to(mel.device))
        return ops.affine_(mel, 4.5, 5.0)              # (melspec + 4.5) / 5

    def extra_repr(self):
        return "winsize={}, hopsize={}".format(self.win_length, self.hopsize)
