"""CPU: the oracle (oracle/maest_oracle.py) against the committed golden fixtures that
oracle/gen_golden.py captured from the IMPORTED reference.  This is what pins the oracle on any
machine where /root/reference is absent (e.g. the GPU box)."""
import os

import numpy as np
import pytest
import torch

from oracle import maest_oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def randn(shape, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy(rng.standard_normal(shape, dtype=np.float32))


def maxdiff(a, b):
    return float((torch.as_tensor(a) - torch.as_tensor(b)).abs().max())


def test_g1_eval_forward_matches_reference_fixture():
    g = np.load(os.path.join(GOLD, "g1_eval_10s.npz"))
    sd = O.make_state_dict(625)
    x = randn((2, 96, 626), 7)
    with torch.no_grad():
        logits, feats = O.forward(x, sd, (96, 625))
        _, emb6 = O.forward(x, sd, (96, 625), transformer_block=6)
    # same op graph as the reference; allow a few ulp for a different BLAS thread count
    assert maxdiff(logits, g["logits"]) < 2e-5
    assert maxdiff(feats, g["features"]) < 2e-5
    assert maxdiff(emb6, g["emb6"]) < 2e-5
    act = torch.sigmoid(logits).mean(0).numpy()
    assert (np.argsort(-act)[:10] == g["top10"]).all()


def test_g4_train_forward_with_captured_draws():
    g = np.load(os.path.join(GOLD, "g4_train_fwd_patchout.npz"))
    sd = O.make_state_dict(625)
    for T in (625, 626):
        x = randn((2, 1, 96, T), 11 + T)
        with torch.no_grad():
            logits, _ = O.forward(x, sd, (96, 625), toffset=int(g[f"toffset_{T}"]),
                                  t_keep=g[f"t_keep_{T}"].tolist())
        assert maxdiff(logits, g[f"logits_{T}"]) < 2e-5
        assert len(g[f"t_keep_{T}"]) == (T - 16) // 10 + 1 - 30


def test_g12_another_patch_stride_matches_reference_fixture():
    """The patch strides are constructor arguments of the reference (get_maest(stride_f=, stride_t=)): (16, 13), evaluation forward and a
    training forward with the reference's own draws (oracle/gen_golden_stride.py)."""
    g = np.load(os.path.join(GOLD, "g12_patch_stride.npz"))
    stride = tuple(int(v) for v in g["stride"])
    sd = O.make_state_dict(625, stride=stride)
    with torch.no_grad():
        logits, feats = O.forward(randn((2, 96, 626), 71), sd, (96, 625), stride=stride)
        lt, _ = O.forward(randn((2, 96, 500), 72), sd, (96, 625), toffset=int(g["toffset"]), t_keep=g["t_keep"].tolist(), stride=stride)
    assert maxdiff(logits, g["logits"]) < 2e-5 and maxdiff(feats, g["features"]) < 2e-5
    assert maxdiff(lt, g["train_logits"]) < 2e-5


def test_g5_training_step_loss_and_grad_probes():
    g = np.load(os.path.join(GOLD, "g5_train_step.npz"))
    sd = {k: v.requires_grad_(True) for k, v in O.make_state_dict(625).items()}
    x = randn((4, 1, 96, 625), 21)
    loss, logits = O.training_loss(x, torch.from_numpy(g["y"]), sd, torch.from_numpy(g["perm"]),
                                   torch.from_numpy(g["lam"]), toffset=int(g["toffset"]),
                                   t_keep=g["t_keep"].tolist())
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) < 1e-6
    names = [n for n, _ in O.state_dict_spec(625, 400)]
    for i, n in enumerate(names):
        if g["grad_present"][i]:
            gn = float(sd[n].grad.norm())
            assert abs(gn - float(g["grad_norm"][i])) <= 1e-4 * float(g["grad_norm"][i]) + 1e-9, n
        else:
            assert sd[n].grad is None, n   # head_dist is unused in "mean" mode


def test_g7_mel_restatement_fixture_and_shapes():
    """Mel parity is UNPINNED at the torchaudio boundary (oracle header); what the reference's tests pin
    is the frame count (tests/test_maest.py:25-43): 10 s -> 626 frames, 30 s -> 1876."""
    g = np.load(os.path.join(GOLD, "g7_mel_restatement_unpinned.npz"))
    fb = O.mel_filterbank()
    assert fb.shape == (257, 96) and int((fb > 0).sum()) == int(g["fb_nnz"]) == 502
    assert np.abs(fb.sum(0) - g["fb_sum"]).max() < 1e-6
    rngw = np.random.Generator(np.random.PCG64(31))
    w = torch.from_numpy((rngw.random((2, 160000), dtype=np.float32) * 2 - 1))
    lm = O.logmel(w)
    assert lm.shape == (2, 96, 626)
    assert maxdiff(lm[:, :, ::25], g["logmel_10s_probe"]) < 1e-4
    assert O.logmel(torch.zeros(1, 480000)).shape == (1, 96, 1876)


def test_oracle_exception_contract():
    sd = O.make_state_dict(625)
    with pytest.raises(Exception):
        O.forward(torch.randn(1, 96, 700), sd, (96, 625))      # 69 patch columns > 62-entry time table
    with pytest.raises(AssertionError):
        O.forward(np.zeros((96, 625)), sd, (96, 625))


def test_mel_restatements_agree_with_an_independent_implementation():
    """torchaudio is absent from this image, so the mel front end (a1) and the kaldi mel banks of AugmentMelSTFT (f3) are
    RESTATEMENTS of torchaudio's published algorithms -- "parity unpinned" against torchaudio's own bits.  What can be checked here: a second,
    independently written implementation of the same published definitions that the image does hold, transformers.audio_utils (Slaney mel
    scale + Slaney area norm, periodic Hann, centred reflect-padded STFT; kaldi mel scale triangularised in mel space)."""
    au = pytest.importorskip("transformers.audio_utils")
    fb = au.mel_filter_bank(num_frequency_bins=257, num_mel_filters=96, min_frequency=0.0, max_frequency=8000.0, sampling_rate=16000,
                            norm="slaney", mel_scale="slaney")
    mine = O.mel_filterbank()
    assert fb.shape == mine.shape == (257, 96) and np.abs(fb - mine).max() < 1e-7 * 32      # (peak weight 0.032; fp32 rounding of the table)
    rng = np.random.Generator(np.random.PCG64(3))
    wave = (rng.standard_normal(16000 * 3) * 0.1).astype(np.float32)
    mel = au.spectrogram(wave.astype(np.float64), window=au.window_function(512, "hann", periodic=True), frame_length=512, hop_length=256,
                         fft_length=512, power=2.0, center=True, pad_mode="reflect", mel_filters=fb, mel_floor=0.0, dtype=np.float64)
    want = (np.log10(1 + mel * 10000) - O.NORM_MEAN) / (O.NORM_STD * 2)
    got = O.logmel(torch.from_numpy(wave)).numpy()
    assert got.shape == want.shape == (96, 1 + wave.size // 256) and np.abs(got - want).max() < 1e-5
    for lo, hi in ((0.0, 16000.0), (20.0, 15000.0), (5.0, 14231.0), (0.0, 15500.0)):          # AugmentMelSTFT's jittered band edges
        kb = au.mel_filter_bank(num_frequency_bins=513, num_mel_filters=128, min_frequency=lo, max_frequency=hi, sampling_rate=32000,
                                norm=None, mel_scale="kaldi", triangularize_in_mel_space=True)
        k = O.kaldi_mel_banks(128, 1024, 32000, lo, hi).numpy()
        assert np.abs(kb.T[:, :512] - k).max() < 5e-5 and np.abs(kb.T[:, 512]).max() == 0.0   # (the oracle forms the mel points in fp32)
