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
