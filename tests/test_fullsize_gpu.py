"""GPU (`-m gpu`): parity at BASELINE.json's FULL sizes (batch 256 x 96 x 626), where the CPU oracle cannot
be run on the whole batch.  The checks are size-independent properties of the path, anchored to the oracle on a
few clips of the very same batch:

  * the clips of a batch are independent and every kernel evaluates a row the same way wherever it sits, so
      - logits of the full batch, restricted to clips S, must EQUAL (bit for bit) the logits of the batch x[S];
      - permuting the batch permutes the outputs (bit for bit);
    the small batch x[S] is then compared with the oracle (fp32 parity mode, 1e-3 relative) -- together: every
    clip of the 256-batch carries the parity of the small one;
  * the training loss is a mean over clips and its gradient is linear in that mean: with mixup and patchout
    draws fixed and mixup partners kept inside each quarter of the batch, the loss / parameter gradients of the
    256-batch equal the average of the four 64-clip steps (fp32 parity mode; the tolerance covers the order of
    the split-K atomics only);
  * the mel front end: the 256-waveform batch equals its per-clip evaluation bit for bit, and the oracle on 2.
"""
import numpy as np
import pytest
import torch

from maest_amd import get_maest
from maest_amd.module import Module
from oracle import maest_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"
B, T = 256, 626


def randn(shape, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy(rng.standard_normal(shape, dtype=np.float32))


def rel_err(a, b):
    a = a.detach().float().cpu()
    b = torch.as_tensor(b).float().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def _model(precision, **kw):
    sd = O.make_state_dict(625, seed=4321)
    m = get_maest("discogs-maest-10s-pw-129e", pretrained=False, precision=precision, **kw)
    m.load_state_dict(sd)
    return m.to(DEV), sd


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_inference_batch256_rows_are_independent_and_match_the_oracle(precision, monkeypatch):
    # bit-exactness needs the SAME GEMM kernel for the 4-clip and the 256-clip batch (below 8192 rows the dispatcher
    # would otherwise pick the 128x128 kernel, whose accumulation order differs in the last bits)
    monkeypatch.setenv("MAEST_GEMM_MIN_M", "512")
    net, sd = _model(precision)
    net.eval()
    x = (0.2 * randn((B, 96, T), 11) + 0.4).to(DEV)          # z-normed log-mel scale
    with torch.no_grad():
        full, feat = net(x)
        assert full.shape == (B, 400) and feat.shape == (B, 768)
        assert torch.isfinite(full).all()
        sel = [0, 1, 77, 255]
        small, sfeat = net(x[sel])
        assert torch.equal(full[sel], small), "a clip's logits must not depend on the batch around it"
        assert torch.equal(feat[sel], sfeat)
        perm = torch.from_numpy(np.random.Generator(np.random.PCG64(3)).permutation(B)).to(DEV)
        pl, _ = net(x[perm])
        assert torch.equal(pl, full[perm]), "permuting the batch must permute the outputs"
    want, wfeat = O.forward(x[sel[:2]].cpu(), sd, (96, 625))
    tol = 1e-3 if precision == "fp32" else 3e-2
    e, ef = rel_err(small[:2], want), rel_err(sfeat[:2], wfeat)
    print(f"full-size inference {precision}: logits rel err {e:.2e}, features {ef:.2e}")
    assert e < tol and ef < tol
    if precision == "fp32":
        assert torch.equal(small[:2].cpu().argsort(dim=1, descending=True)[:, :10],
                           want.argsort(dim=1, descending=True)[:, :10]), "top-10 label indices must be identical"


def test_training_step_batch256_is_the_mean_of_its_quarters_fp32():
    net, _ = _model("fp32", input_t=625, s_patchout_t=30)
    net.train()
    mod = Module(net=net, mixup_alpha=0.3)
    x = randn((B, 1, 96, T), 21).to(DEV)
    rng = np.random.Generator(np.random.PCG64(22))
    y = torch.from_numpy((rng.random((B, 400)) < 0.00625).astype(np.float32)).to(DEV)
    Q = B // 4
    perm = torch.cat([torch.from_numpy(rng.permutation(Q)) + q * Q for q in range(4)])       # partners stay in-quarter
    lam = torch.from_numpy(np.maximum(b := rng.beta(0.3, 0.3, B).astype(np.float32), 1 - b))
    Tp = (T - 16) // 10 + 1
    keep = torch.from_numpy(np.sort(rng.permutation(Tp)[: Tp - 30]))
    po = (3, keep)
    names = ["blocks.0.attn.qkv.weight", "blocks.11.mlp.fc2.weight", "blocks.5.norm1.weight", "patch_embed.proj.weight",
             "time_new_pos_embed", "head.1.bias", "blocks.7.attn.proj.bias"]
    params = dict(net.named_parameters())

    def step(xs, ys, mix):
        for p in net.parameters():
            p.grad = None
        loss = mod.training_step((xs, None, ys), 0, _mixup=mix, _patchout=po)
        loss.backward()
        return loss.item(), {n: params[n].grad.detach().clone() for n in names}

    loss_full, g_full = step(x, y, (perm, lam))
    acc_loss, acc = 0.0, {n: torch.zeros_like(g_full[n]) for n in names}
    for q in range(4):
        s = slice(q * Q, (q + 1) * Q)
        lq, gq = step(x[s], y[s], (perm[s] - q * Q, lam[s]))
        acc_loss += lq / 4
        for n in names:
            acc[n] += gq[n] / 4
    assert abs(loss_full - acc_loss) <= 2e-6 * abs(acc_loss), (loss_full, acc_loss)
    for n in names:
        e = rel_err(g_full[n], acc[n])
        assert e < 2e-4, f"{n}: full-batch gradient differs from the mean of the quarter steps by {e:.2e}"


def test_mel_frontend_batch256_waveforms():
    from maest_amd.melspectrogram import MelSpectrogram
    rng = np.random.Generator(np.random.PCG64(31))
    w = torch.from_numpy((rng.random((B, 160000), dtype=np.float32) * 2 - 1) * 0.5)
    mel = MelSpectrogram()
    got = mel(w.to(DEV))
    assert got.shape == (B, 96, 626)
    for i in (0, 100, 255):
        assert torch.equal(got[i], mel(w[i:i + 1].to(DEV))[0]), "a clip's log-mel must not depend on the batch"
    want = O.logmel(w[:2])
    assert (got[:2].cpu() - want).abs().max().item() < 2e-4
