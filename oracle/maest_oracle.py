"""CPU oracle for the MAEST mel -> patchout-ViT hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch (CPU, fp32) restatement of the reference algorithm.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it; nothing under ``maest_amd/`` does (the product path fails
loudly when the HIP library is missing instead of falling back to this).

Pinning status
--------------
* ViT path (patch embed .. head, training loss and gradients): PINNED.  Checked
  bit-for-bit against the imported reference (``/root/reference/models/maest.py``)
  by ``oracle/gen_golden.py`` and against the committed fixtures in
  ``tests/golden/`` by ``tests/test_oracle_golden.py``.
* Mel front end: PARITY UNPINNED.  The reference delegates the arithmetic to
  ``torchaudio.transforms.{Spectrogram, MelScale}`` (``models/helpers/melspectrogram.py:3,29-42``;
  ``pyproject.toml`` declares ``torchaudio`` without a version pin and the package is
  absent from this image).  The restatement below follows torchaudio's published
  algorithm (``torch.stft`` center/reflect, periodic Hann, ``melscale_fbanks`` with
  slaney scale + slaney norm); the only things the reference's own tests pin for this
  path are frame counts / shapes (``tests/test_maest.py:25-43``), which are checked.
  Cross-check (not a pin: it is not torchaudio): the filter bank, the log-mel of a waveform and the
  kaldi banks of ``augment_mel`` agree with ``transformers.audio_utils`` -- an independent implementation
  of the same published definitions that this image holds -- to 1e-9 / 8e-7 / 2e-5
  (``tests/test_oracle_golden.py::test_mel_restatements_agree_with_an_independent_implementation``).
* Patch strides other than (10, 10): PINNED at (16, 13) (``oracle/gen_golden_stride.py``, fixture g12).

Every function cites the reference lines it follows (paths relative to /root/reference).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------
# constants (models/helpers/melspectrogram.py:16-24)
# --------------------------------------------------------------------------------------
SR = 16000
WIN_LEN = 512
HOP_LEN = 256
N_MEL = 96
NORM_MEAN = 2.06755686098554
NORM_STD = 1.268292820667291

EMBED_DIM = 768
DEPTH = 12
NUM_HEADS = 12
MLP_HIDDEN = 3072
PATCH = 16


# --------------------------------------------------------------------------------------
# mel front end  (models/helpers/melspectrogram.py:13-60)
# --------------------------------------------------------------------------------------
def _hz_to_mel_slaney(f: np.ndarray) -> np.ndarray:
    # torchaudio.functional._hz_to_mel(mel_scale="slaney"): linear below 1 kHz
    # (200/3 Hz per mel), logarithmic above with step log(6.4)/27.
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = math.log(6.4) / 27.0
    with np.errstate(divide="ignore", invalid="ignore"):
        log_part = min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep
    return np.where(f >= min_log_hz, log_part, mels)


def _mel_to_hz_slaney(m: np.ndarray) -> np.ndarray:
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    freqs = f_sp * m
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = math.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), freqs)


def mel_filterbank(n_freqs: int = WIN_LEN // 2 + 1, f_min: float = 0.0, f_max: float = SR / 2,
                   n_mels: int = N_MEL, sample_rate: int = SR) -> np.ndarray:
    """``torchaudio.functional.melscale_fbanks(n_freqs, f_min, f_max, n_mels, sample_rate,
    norm="slaney", mel_scale="slaney")`` -> float32 ``[n_freqs, n_mels]``.

    Reference call site: ``MelScale(n_mels=96, sample_rate=16000, n_stft=257, norm="slaney",
    mel_scale="slaney")`` (melspectrogram.py:36-42); MelScale defaults f_min=0, f_max=sr//2.
    torchaudio computes this table in float32; we compute in float64 and round once, which
    differs from torchaudio's own float32 table by at most ~1 ulp per weight.
    """
    all_freqs = np.linspace(0, sample_rate // 2, n_freqs)
    m_min = _hz_to_mel_slaney(np.array(f_min))
    m_max = _hz_to_mel_slaney(np.array(f_max))
    m_pts = np.linspace(m_min, m_max, n_mels + 2)
    f_pts = _mel_to_hz_slaney(m_pts)
    f_diff = f_pts[1:] - f_pts[:-1]                       # [n_mels+1]
    slopes = f_pts[None, :] - all_freqs[:, None]          # [n_freqs, n_mels+2]
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    fb = np.maximum(0.0, np.minimum(down, up))
    enorm = 2.0 / (f_pts[2:n_mels + 2] - f_pts[:n_mels])  # slaney area norm
    fb = fb * enorm[None, :]
    return fb.astype(np.float32)


def power_spectrogram(wave: torch.Tensor) -> torch.Tensor:
    """``Spectrogram(n_fft=512, win_length=512, hop_length=256, power=2)``
    (melspectrogram.py:29-34,49): center=True reflect pad, periodic Hann, onesided,
    not normalized.  ``[..., S] -> [..., 257, 1 + S // 256]``."""
    win = torch.hann_window(WIN_LEN, periodic=True, dtype=wave.dtype)
    shape = wave.shape
    x = wave.reshape(-1, shape[-1])
    st = torch.stft(x, n_fft=WIN_LEN, hop_length=HOP_LEN, win_length=WIN_LEN, window=win,
                    center=True, pad_mode="reflect", normalized=False, onesided=True,
                    return_complex=True)
    p = st.real ** 2 + st.imag ** 2
    return p.reshape(shape[:-1] + p.shape[-2:])


def logmel(wave: torch.Tensor) -> torch.Tensor:
    """``MelSpectrogram.forward`` (melspectrogram.py:47-60): power spec -> mel ->
    ``log10(1 + 1e4 * mel)`` -> z-norm ``(x - mean) / (2 * std)``."""
    spec = power_spectrogram(wave.float())
    fb = torch.from_numpy(mel_filterbank())
    mel = torch.matmul(spec.transpose(-1, -2), fb).transpose(-1, -2)
    lm = torch.log10(1 + mel * 10000)
    return (lm - NORM_MEAN) / (NORM_STD * 2)


# --------------------------------------------------------------------------------------
# architecture registry (models/maest.py:1151-1388, 1509-1530)
# --------------------------------------------------------------------------------------
ARCH_DEFAULT_T = {
    "passt_deit_bd_p16_384": 998,
    "passt_s_swa_p16_128_ap476": 998,
    "discogs-maest-10s-fs-129e": 625,
    "discogs-maest-10s-pw-129e": 625,
    "discogs-maest-10s-dw-75e": 625,
    "discogs-maest-5s-pw-129e": 312,
    "discogs-maest-20s-pw-129e": 1250,
    "discogs-maest-30s-pw-129e": 1875,
    "discogs-maest-30s-pw-73e-ts": 1875,
    "discogs-maest-30s-pw-129e-519l": 1875,
}


def state_dict_spec(img_t: int, n_classes: int = 400, img_f: int = 96, stride=10):
    """Ordered (name, shape) list of the MAEST state_dict (order is the
    reference's ``state_dict()`` order, probed).  stride: int or (frequency, time)."""
    D = EMBED_DIM
    sf, st = (stride, stride) if isinstance(stride, int) else stride
    gf, gt = img_f // sf, img_t // st               # PatchEmbed.grid_size (maest.py:234)
    spec = [
        ("cls_token", (1, 1, D)), ("dist_token", (1, 1, D)), ("new_pos_embed", (1, 2, D)),
        ("freq_new_pos_embed", (1, D, gf, 1)), ("time_new_pos_embed", (1, D, 1, gt)),
        ("patch_embed.proj.weight", (D, 1, PATCH, PATCH)), ("patch_embed.proj.bias", (D,)),
    ]
    for i in range(DEPTH):
        p = f"blocks.{i}."
        spec += [
            (p + "norm1.weight", (D,)), (p + "norm1.bias", (D,)),
            (p + "attn.qkv.weight", (3 * D, D)), (p + "attn.qkv.bias", (3 * D,)),
            (p + "attn.proj.weight", (D, D)), (p + "attn.proj.bias", (D,)),
            (p + "norm2.weight", (D,)), (p + "norm2.bias", (D,)),
            (p + "mlp.fc1.weight", (MLP_HIDDEN, D)), (p + "mlp.fc1.bias", (MLP_HIDDEN,)),
            (p + "mlp.fc2.weight", (D, MLP_HIDDEN)), (p + "mlp.fc2.bias", (D,)),
        ]
    spec += [
        ("norm.weight", (D,)), ("norm.bias", (D,)),
        ("head.0.weight", (D,)), ("head.0.bias", (D,)),
        ("head.1.weight", (n_classes, D)), ("head.1.bias", (n_classes,)),
        ("head_dist.weight", (n_classes, D)), ("head_dist.bias", (n_classes,)),
    ]
    return spec


def make_state_dict(img_t: int, n_classes: int = 400, seed: int = 1234,
                    std: float = 0.02, stride=10) -> Dict[str, torch.Tensor]:
    """Deterministic synthetic weights: ``numpy.random.Generator(PCG64(seed))`` filling the
    state_dict in key order: N(0, std^2) everywhere, LayerNorm gains
    1 + N(0, std^2).  Independent of torch's RNG so the GPU box regenerates the same
    tensors that the golden fixtures were captured with."""
    rng = np.random.Generator(np.random.PCG64(seed))
    sd = {}
    for name, shape in state_dict_spec(img_t, n_classes, stride=stride):
        a = rng.standard_normal(shape, dtype=np.float32) * np.float32(std)
        if (".norm" in name or name.startswith("norm.") or name.startswith("head.0.")) \
                and name.endswith("weight"):
            a = a + np.float32(1.0)
        sd[name] = torch.from_numpy(a)
    return sd


# --------------------------------------------------------------------------------------
# input rank dispatch (models/maest.py:855-895)
# --------------------------------------------------------------------------------------
def prepare_input(x: torch.Tensor, img_size: Tuple[int, int],
                  melspectrogram_input: bool = False) -> torch.Tensor:
    """-> ``[B, 1, F, T]``.  Mirrors maest.py:855-895 (without the in-place unsqueeze_)."""
    assert isinstance(x, torch.Tensor), "Input must be a torch.Tensor"
    assert x.nelement() > 0, "Input tensor must not be empty"
    if x.dim() == 1:
        assert melspectrogram_input is False
        x = logmel(x)
        if x.shape[1] >= img_size[1]:
            trim = x.shape[1] % img_size[1]
            if trim:
                x = x[:, :-trim]
            x = x.reshape(img_size[0], 1, -1, img_size[1])
            x = torch.swapaxes(x, 0, 2)
        else:
            x = x.reshape(1, 1, x.shape[0], x.shape[1])
    elif x.dim() == 2 and melspectrogram_input:
        trim = x.shape[1] % img_size[1]
        if trim:
            x = x[:, :-trim]
        x = x.reshape(img_size[0], 1, -1, img_size[1])
        x = torch.swapaxes(x, 0, 2)
    elif x.dim() == 2:
        x = logmel(x).unsqueeze(1)
    elif x.dim() == 3:
        x = x.unsqueeze(1)
    return x


# --------------------------------------------------------------------------------------
# ViT pieces
# --------------------------------------------------------------------------------------
def patch_embed(x: torch.Tensor, sd, stride=10) -> torch.Tensor:
    """``PatchEmbed.forward`` with flatten=False (maest.py:243-256, 506-513):
    Conv2d(1, 768, k=16, s=10) -> ``[B, 768, 9, T']``; the stride is a constructor argument (maest.py:214-241; every published
    architecture uses 10)."""
    return F.conv2d(x, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], stride=stride)


def tokens_from_patches(p: torch.Tensor, sd, toffset: int = 0,
                        t_keep: Optional[Sequence[int]] = None,
                        f_keep: Optional[Sequence[int]] = None,
                        u_keep: Optional[Sequence[int]] = None) -> torch.Tensor:
    """Positional add + structured patchout + flatten + cls/dist concat
    (maest.py:645-675, 678-701, 769, 785-796).  ``toffset`` / ``t_keep`` / ``f_keep`` are the
    values the reference draws from ``torch.randint`` / ``torch.randperm`` in training mode
    (eval: toffset=0, keep everything)."""
    B, E, Fd, Td = p.shape
    tpe = sd["time_new_pos_embed"]
    if Td > tpe.shape[-1]:
        raise Exception(
            f"the patches shape:{p.shape} are larger than the expected time encodings {tpe.shape},"
            " please reduce the input duration.")
    p = p + tpe[:, :, :, toffset:toffset + Td]
    p = p + sd["freq_new_pos_embed"]
    if t_keep is not None:
        p = p[:, :, :, torch.as_tensor(list(t_keep), dtype=torch.long)]
    if f_keep is not None:
        p = p[:, :, torch.as_tensor(list(f_keep), dtype=torch.long), :]
    x = p.flatten(2).transpose(1, 2)
    if u_keep is not None:      # unstructured patchout: sorted random subset of the sequence (maest.py:773-780)
        x = x[:, torch.as_tensor(list(u_keep), dtype=torch.long), :]
    cls = sd["cls_token"].expand(B, -1, -1) + sd["new_pos_embed"][:, :1, :]
    dist = sd["dist_token"].expand(B, -1, -1) + sd["new_pos_embed"][:, 1:, :]
    return torch.cat((cls, dist, x), dim=1)


def attention(x: torch.Tensor, sd, pre: str) -> torch.Tensor:
    """``Attention.forward`` (maest.py:358-378)."""
    B, N, C = x.shape
    H = NUM_HEADS
    qkv = F.linear(x, sd[pre + "qkv.weight"], sd[pre + "qkv.bias"])
    qkv = qkv.reshape(B, N, 3, H, C // H).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    attn = (q @ k.transpose(-2, -1)) * ((C // H) ** -0.5)
    attn = attn.softmax(dim=-1)
    y = (attn @ v).transpose(1, 2).reshape(B, N, C)
    return F.linear(y, sd[pre + "proj.weight"], sd[pre + "proj.bias"])


def mlp(x: torch.Tensor, sd, pre: str) -> torch.Tensor:
    """``Mlp.forward`` (maest.py:202-208); exact-erf GELU (maest.py:500)."""
    h = F.gelu(F.linear(x, sd[pre + "fc1.weight"], sd[pre + "fc1.bias"]))
    return F.linear(h, sd[pre + "fc2.weight"], sd[pre + "fc2.bias"])


def block(x: torch.Tensor, sd, i: int, return_self_attention: bool = False) -> torch.Tensor:
    """``Block.forward`` (maest.py:414-420); LayerNorm eps 1e-6 (maest.py:499)."""
    p = f"blocks.{i}."
    h = F.layer_norm(x, (EMBED_DIM,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-6)
    a = attention(h, sd, p + "attn.")
    if return_self_attention:
        return a
    x = x + a
    h = F.layer_norm(x, (EMBED_DIM,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-6)
    return x + mlp(h, sd, p + "mlp.")


def forward_features(x4: torch.Tensor, sd, transformer_block: int = -1,
                     return_self_attention: bool = False, toffset: int = 0,
                     t_keep=None, f_keep=None, probes: Optional[list] = None, u_keep=None, stride=10):
    """``MAEST.forward_features`` (maest.py:634-829)."""
    x = tokens_from_patches(patch_embed(x4, sd, stride), sd, toffset, t_keep, f_keep, u_keep)
    if transformer_block == -1:
        for i in range(DEPTH):
            x = block(x, sd, i)
            if probes is not None:
                probes.append(x)
        x = F.layer_norm(x, (EMBED_DIM,), sd["norm.weight"], sd["norm.bias"], 1e-6)
        return x[:, 0], x[:, 1]
    for i in range(DEPTH):
        if i == transformer_block:
            x = block(x, sd, i, return_self_attention=return_self_attention)
            break
        x = block(x, sd, i)
    return torch.cat([x[:, 0, :], x[:, 1, :], torch.mean(x[:, 2:, :], dim=1)], dim=1)


def forward(x: torch.Tensor, sd, img_size: Tuple[int, int], transformer_block: int = -1,
            return_self_attention: bool = False, melspectrogram_input: bool = False,
            distilled_type: str = "mean", toffset: int = 0, t_keep=None, f_keep=None,
            probes: Optional[list] = None, u_keep=None, stride=10):
    """``MAEST.forward`` (maest.py:831-933)."""
    x4 = prepare_input(x, img_size, melspectrogram_input)
    out = forward_features(x4, sd, transformer_block, return_self_attention,
                           toffset, t_keep, f_keep, probes, u_keep, stride)
    if transformer_block != -1:
        return None, out
    cls, dist = out
    features = (cls + dist) / 2
    def head(z):
        z = F.layer_norm(z, (EMBED_DIM,), sd["head.0.weight"], sd["head.0.bias"], 1e-5)
        return F.linear(z, sd["head.1.weight"], sd["head.1.bias"])
    if distilled_type == "mean":
        return head(features), features
    if distilled_type == "separated":
        return head(cls), F.linear(dist, sd["head_dist.weight"], sd["head_dist.bias"]), features
    raise ValueError(distilled_type)


def predict_labels(x: torch.Tensor, sd, img_size) -> np.ndarray:
    """``MAEST.predict_labels`` (maest.py:935-939), activations only."""
    logits = forward(x, sd, img_size)[0]
    return torch.sigmoid(logits).mean(dim=0).detach().cpu().numpy()


# --------------------------------------------------------------------------------------
# training step (models/module.py:73-102, 280-316; helpers/mixup.py:5-12)
# --------------------------------------------------------------------------------------
def mixup(x: torch.Tensor, perm: torch.Tensor, lam: torch.Tensor) -> torch.Tensor:
    """``x * lam + x[perm] * (1 - lam)`` with lam broadcast over the batch axis
    (module.py:77-86)."""
    shape = (x.shape[0],) + (1,) * (x.dim() - 1)
    return x * lam.reshape(shape) + x[perm] * (1.0 - lam.reshape(shape))


def training_loss(x: torch.Tensor, y: torch.Tensor, sd, perm=None, lam=None, toffset: int = 0,
                  t_keep=None, f_keep=None, y_teacher: Optional[torch.Tensor] = None, stride=10):
    """``Module.training_step`` / ``TeacherStudentModule.training_step`` given the drawn
    (perm, lam, toffset, t_keep).  Returns (loss, logits...)."""
    if perm is not None:
        x = mixup(x, perm, lam)
        y = mixup(y, perm, lam)
        if y_teacher is not None:
            y_teacher = mixup(y_teacher, perm, lam)
    img = (x.shape[-2], x.shape[-1])
    if y_teacher is None:
        logits, _ = forward(x, sd, img, toffset=toffset, t_keep=t_keep, f_keep=f_keep, stride=stride)
        return F.binary_cross_entropy_with_logits(logits, y), logits
    lc, ld, _ = forward(x, sd, img, distilled_type="separated", toffset=toffset,
                        t_keep=t_keep, f_keep=f_keep, stride=stride)
    loss = (F.binary_cross_entropy_with_logits(lc, y)
            + F.binary_cross_entropy_with_logits(ld, y_teacher)) / 2
    return loss, lc, ld


def spec_masking(x: torch.Tensor, time_masks: Sequence[Tuple[int, int]],
                 freq_masks: Sequence[Tuple[int, int]]) -> torch.Tensor:
    """``SpecMasking.compute`` (helpers/spec_masking.py:27-33) on explicit integer
    (start, width) lists: fill 0.0 on ``[start, start+width)`` along time then frequency.
    The sampling of (start, width) is torchaudio's (parity unpinned)."""
    x = x.clone()
    for s, w in time_masks:
        x[..., :, s:s + w] = 0.0
    for s, w in freq_masks:
        x[..., s:s + w, :] = 0.0
    return x


# ----------------------------------------------------------------------------------------------
# AugmentMelSTFT (models/preprocess.py:17-128), eval-mode path and explicit-stripe training path.
# PARITY UNPINNED: torch.stft is torch's own, but the mel banks come from torchaudio.compliance.kaldi
# (absent here); `kaldi_mel_banks` restates its published algorithm (VTLN warp factor 1.0).
# ----------------------------------------------------------------------------------------------
def kaldi_mel_banks(num_bins: int, n_fft: int, sr: float, low_freq: float, high_freq: float) -> torch.Tensor:
    nyquist = 0.5 * sr
    if high_freq <= 0.0:
        high_freq += nyquist
    mel = lambda f: 1127.0 * torch.log(1.0 + f / 700.0)
    mel_low = 1127.0 * math.log(1.0 + low_freq / 700.0)
    mel_high = 1127.0 * math.log(1.0 + high_freq / 700.0)
    delta = (mel_high - mel_low) / (num_bins + 1)
    b = torch.arange(num_bins).unsqueeze(1)
    left, center, right = mel_low + b * delta, mel_low + (b + 1.0) * delta, mel_low + (b + 2.0) * delta
    m = mel((sr / n_fft) * torch.arange(n_fft // 2)).unsqueeze(0)
    return torch.max(torch.zeros(1), torch.min((m - left) / (center - left), (right - m) / (right - center)))


def augment_mel(x: torch.Tensor, n_mels=128, sr=32000, win_length=800, hopsize=320, n_fft=1024, fmin=0.0,
                fmax=15500.0, f_stripe=None, t_stripe=None) -> torch.Tensor:
    """preprocess.py:81-131 for given band edges; `f_stripe` / `t_stripe` = (start, width) zeroed after the log
    (the training-time masks, drawn by the caller)."""
    x = F.conv1d(x.unsqueeze(1), torch.as_tensor([[[-0.97, 1.0]]])).squeeze(1)
    spec = torch.stft(x, n_fft, hop_length=hopsize, win_length=win_length, center=True, normalized=False,
                      window=torch.hann_window(win_length, periodic=False), return_complex=True)
    power = spec.real ** 2 + spec.imag ** 2
    basis = F.pad(kaldi_mel_banks(n_mels, n_fft, sr, fmin, fmax), (0, 1), mode="constant", value=0)
    mel = (torch.matmul(basis, power) + 0.00001).log()
    if f_stripe is not None:
        mel[:, f_stripe[0]:f_stripe[0] + f_stripe[1], :] = 0.0
    if t_stripe is not None:
        mel[:, :, t_stripe[0]:t_stripe[0] + t_stripe[1]] = 0.0
    return (mel + 4.5) / 5.0
