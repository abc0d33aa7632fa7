"""On-the-fly waveform -> log-mel front end, same module surface as the reference's
``models/helpers/melspectrogram.py:13-60`` (class constants :16-24, ``znorm`` :44-45,
``forward`` :47-60), with the arithmetic done by ONE fused HIP kernel (csrc/mel.hip) instead of
torchaudio's Spectrogram + MelScale + three elementwise passes.

The filterbank / window / twiddle tables are computed once on the host in float64 (slaney mel scale,
slaney area normalisation, f in [0, sr/2] -- torchaudio ``melscale_fbanks`` semantics) and shipped to
the device in band-sparse form: each triangular band only touches a short run of FFT bins.
"""
from __future__ import annotations

import math

import numpy as np
import torch
from torch.nn import Module

from . import ops


def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = math.log(6.4) / 27.0
    out = f / f_sp
    hi = f >= min_log_hz
    out = np.where(hi, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, out)
    return out


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = math.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def slaney_filterbank(n_freqs=257, n_mels=96, sample_rate=16000, f_min=0.0, f_max=None):
    """float32 [n_freqs, n_mels] triangular filters, slaney scale, slaney norm."""
    f_max = sample_rate / 2 if f_max is None else f_max
    all_freqs = np.linspace(0, sample_rate // 2, n_freqs)
    m_pts = np.linspace(_hz_to_mel(f_min), _hz_to_mel(f_max), n_mels + 2)
    f_pts = _mel_to_hz(m_pts)
    f_diff = np.diff(f_pts)
    slopes = f_pts[None, :] - all_freqs[:, None]
    down = -slopes[:, :-2] / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    fb = np.maximum(0.0, np.minimum(down, up))
    fb *= (2.0 / (f_pts[2:] - f_pts[:-2]))[None, :]
    return fb.astype(np.float32)


class MelConstants:
    """Device-resident tables for csrc/mel.hip."""

    def __init__(self, device, sr=16000, n_fft=512, n_mel=96, norm_mean=0.0, norm_std=0.5, log_scale=10000.0):
        fb = slaney_filterbank(n_fft // 2 + 1, n_mel, sr)          # [257, 96]
        starts, lens = [], []
        for m in range(n_mel):
            nz = np.nonzero(fb[:, m])[0]
            if len(nz) == 0:
                starts.append(0)
                lens.append(0)
            else:
                starts.append(int(nz[0]))
                lens.append(int(nz[-1] - nz[0] + 1))
        self.fb_stride = max(8, int(math.ceil(max(lens) / 8) * 8))
        w = np.zeros((n_mel, self.fb_stride), np.float32)
        for m in range(n_mel):
            w[m, : lens[m]] = fb[starts[m]: starts[m] + lens[m], m]
        n = np.arange(n_fft, dtype=np.float64)
        window = (0.5 - 0.5 * np.cos(2.0 * np.pi * n / n_fft)).astype(np.float32)   # periodic Hann
        ang = -2.0 * np.pi * n / n_fft
        tw = np.stack([np.cos(ang), np.sin(ang)], axis=1).astype(np.float32)       # [512, 2]
        self.window = torch.from_numpy(window).to(device)
        self.twiddle = torch.from_numpy(tw).contiguous().to(device)
        self.fb_start = torch.tensor(starts, dtype=torch.int32, device=device)
        self.fb_len = torch.tensor(lens, dtype=torch.int32, device=device)
        self.fb_w = torch.from_numpy(w).to(device)
        self.log_scale = float(log_scale)
        self.norm_mean = float(norm_mean)
        self.norm_2std = float(norm_std * 2)


class _ConstantBuffers(Module):
    """Holds the persistent buffers torchaudio's transforms register, so that state_dict keys / shapes equal the
    reference's (``melspectrogram.spec.window`` [512] from torchaudio Spectrogram, ``melspectrogram.mel_scale.fb``
    [257, 96] from MelScale).  They are constants of the algorithm: a state_dict that carries them loads cleanly
    (strict), one that lacks them (a checkpoint written with the mel module excluded, the oracle's spec) does too,
    and the values in a checkpoint never override the tables the kernel uses."""

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                              error_msgs):
        for name, buf in self._buffers.items():
            v = state_dict.get(prefix + name)
            if v is not None and tuple(v.shape) != tuple(buf.shape):
                error_msgs.append(f"size mismatch for {prefix + name}: checkpoint {tuple(v.shape)}, "
                                  f"model {tuple(buf.shape)}")


class MelSpectrogram(Module):
    """Extract z-normalised log-mel spectrograms (drop-in for the reference module)."""

    sr = 16000
    win_len = 512
    hop_len = 256
    power = 2
    n_mel = 96
    norm = "slaney"
    mel_scale_type = "slaney"
    norm_mean = 2.06755686098554
    norm_std = 1.268292820667291

    def __init__(self):
        super().__init__()
        self._consts = {}
        n = np.arange(self.win_len, dtype=np.float64)
        self.spec = _ConstantBuffers()                   # reference: torchaudio Spectrogram (melspectrogram.py:29-34)
        self.spec.register_buffer("window", torch.from_numpy(
            (0.5 - 0.5 * np.cos(2.0 * np.pi * n / self.win_len)).astype(np.float32)))
        self.mel_scale = _ConstantBuffers()              # reference: torchaudio MelScale (melspectrogram.py:36-42)
        self.mel_scale.register_buffer("fb", torch.from_numpy(
            slaney_filterbank(self.win_len // 2 + 1, self.n_mel, self.sr)))

    def _constants(self, device):
        key = str(device)
        if key not in self._consts:
            self._consts[key] = MelConstants(device, self.sr, self.win_len, self.n_mel, self.norm_mean,
                                             self.norm_std)
        return self._consts[key]

    def znorm(self, input_values: torch.Tensor) -> torch.Tensor:
        return (input_values - (self.norm_mean)) / (self.norm_std * 2)

    def forward(self, waveform: torch.Tensor) -> torch.Tensor:
        """[S] -> [96, T]   or   [B, S] -> [B, 96, T]   with T = 1 + S // 256."""
        squeeze = waveform.dim() == 1
        w = waveform.reshape(-1, waveform.shape[-1])
        if w.dtype != torch.float32:
            w = w.float()
        w = w.contiguous()
        out = ops.logmel(w, self._constants(w.device))
        if squeeze:
            return out[0]
        return out.reshape(waveform.shape[:-1] + out.shape[-2:])
