#!/usr/bin/env python
"""Generate tests/golden/*.npz from the IMPORTED reference (run in the authoring container only).

TEST INFRASTRUCTURE.  Imports ``/root/reference/models/maest.py`` with three ``sys.modules``
stubs (sacred, timm.models.load_pretrained, torchaudio.transforms -- SURVEY 8c), drives it
with deterministic synthetic weights/inputs (``oracle.maest_oracle.make_state_dict`` and
PCG64-seeded inputs, so nothing but OUTPUTS needs to be stored), asserts that the oracle
restatement reproduces the reference, and writes the outputs as small fixtures.

    python oracle/gen_golden.py            # writes tests/golden/*.npz

The reference cannot travel to the GPU box; the fixtures + this script are what is committed.
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import maest_oracle as O  # noqa: E402

REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")


def import_reference():
    sacred = types.ModuleType("sacred")

    class Ingredient:
        def __init__(self, *a, **k):
            pass

        def config(self, f):
            return f

        def capture(self, f):
            return f

        def command(self, f):
            return f

    sacred.Ingredient = Ingredient
    sys.modules["sacred"] = sacred
    timm = types.ModuleType("timm")
    timm_models = types.ModuleType("timm.models")

    def load_pretrained(*a, **k):
        raise RuntimeError("no network")

    timm_models.load_pretrained = load_pretrained
    timm.models = timm_models
    sys.modules["timm"] = timm
    sys.modules["timm.models"] = timm_models
    ta = types.ModuleType("torchaudio")
    tat = types.ModuleType("torchaudio.transforms")

    class _M(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    tat.Spectrogram = _M
    tat.MelScale = _M
    ta.transforms = tat
    sys.modules["torchaudio"] = ta
    sys.modules["torchaudio.transforms"] = tat
    sys.path.insert(0, REF)
    import models.maest as rm
    return rm


def randn(shape, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy(rng.standard_normal(shape, dtype=np.float32))


def build(rm, arch, img_t, n_classes=400, seed=1234, **kw):
    m = rm.get_maest(arch, pretrained=False, n_classes=n_classes, **kw)
    sd = O.make_state_dict(img_t, n_classes=m.num_classes, seed=seed)
    missing = m.load_state_dict(sd, strict=True)
    return m, sd


def check(name, a, b, tol=0.0):
    d = (a - b).abs().max().item()
    print(f"  [{name}] oracle-vs-reference max|diff| = {d:.3e}")
    assert d <= tol, (name, d)


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    rm = import_reference()

    # ---------------- G1: eval forward, 10 s, N = 560 ------------------------------------
    print("G1")
    m, sd = build(rm, "discogs-maest-10s-pw-129e", 625)
    m.eval()
    x = randn((2, 96, 626), 7)
    with torch.no_grad():
        logits, feats = m(x.clone())
        _, emb6 = m(x.clone(), transformer_block=6)
        _, att3 = m(x.clone(), transformer_block=3, return_self_attention=True)
        probes = []
        ol, of = O.forward(x.clone(), sd, (96, 625), probes=probes)
        _, oe6 = O.forward(x.clone(), sd, (96, 625), transformer_block=6)
        _, oa3 = O.forward(x.clone(), sd, (96, 625), transformer_block=3, return_self_attention=True)
    check("G1 logits", ol, logits)
    check("G1 feats", of, feats)
    check("G1 emb6", oe6, emb6)
    check("G1 att3", oa3, att3)
    # per-block probes come from the (now validated) oracle; localise failures on the GPU
    blk_probe = torch.stack([p[:, :2, :8] for p in probes]).numpy()
    blk_norm = torch.stack([p.norm(dim=(1, 2)) for p in probes]).numpy()
    act = torch.sigmoid(logits).mean(0).numpy()
    np.savez(os.path.join(OUT, "g1_eval_10s.npz"), logits=logits.numpy(), features=feats.numpy(),
             emb6=emb6.numpy(), att3=att3.numpy(), blk_probe=blk_probe, blk_norm=blk_norm,
             activations=act, top10=np.argsort(-act)[:10])

    # second input scale: mel-like statistics
    x2 = 0.2 * randn((2, 96, 626), 8) + 0.4
    with torch.no_grad():
        l2, f2 = m(x2.clone())
        ol2, of2 = O.forward(x2.clone(), sd, (96, 625))
    check("G1b logits", ol2, l2)
    np.savez(os.path.join(OUT, "g1b_eval_10s_mellike.npz"), logits=l2.numpy(), features=f2.numpy())

    # ---------------- G2: 30 s / 519 labels, N = 1685, and the chunking path --------------
    print("G2")
    m30, sd30 = build(rm, "discogs-maest-30s-pw-129e-519l", 1875)
    m30.eval()
    x30 = randn((1, 96, 1876), 9)
    xc = randn((96, 3752), 10)            # 2-D mel -> 2 chunks of 1875 (trim 2)
    with torch.no_grad():
        l30, f30 = m30(x30.clone())
        lc, fc = m30(xc.clone(), melspectrogram_input=True)
        ol30, of30 = O.forward(x30.clone(), sd30, (96, 1875))
        olc, ofc = O.forward(xc.clone(), sd30, (96, 1875), melspectrogram_input=True)
    check("G2 logits", ol30, l30)
    check("G2 chunk logits", olc, lc)
    np.savez(os.path.join(OUT, "g2_eval_30s_519.npz"), logits=l30.numpy(), features=f30.numpy(),
             chunk_logits=lc.numpy(), chunk_features=fc.numpy())

    # ---------------- G4: train-mode forward with captured patchout draws ------------------
    print("G4")
    g4 = {}
    for T in (625, 626):
        mt, sdt = build(rm, "passt_s_swa_p16_128_ap476", 625, input_t=625, s_patchout_t=30)
        mt.train()
        xt = randn((2, 1, 96, T), 11 + T)
        Tp = (T - 16) // 10 + 1
        table = 62
        # replay the reference's RNG draws (maest.py:648-650, 684-686) to capture them
        torch.manual_seed(100 + T)
        toff = torch.randint(1 + table - Tp, (1,)).item()
        keep = torch.randperm(Tp)[: Tp - 30].sort().values
        torch.manual_seed(100 + T)
        with torch.no_grad():
            lt, ft = mt(xt.clone())
            olt, oft = O.forward(xt.clone(), sdt, (96, 625), toffset=toff, t_keep=keep.tolist())
        check(f"G4 T={T} logits", olt, lt)
        g4[f"logits_{T}"] = lt.numpy()
        g4[f"features_{T}"] = ft.numpy()
        g4[f"toffset_{T}"] = np.int64(toff)
        g4[f"t_keep_{T}"] = keep.numpy()
    np.savez(os.path.join(OUT, "g4_train_fwd_patchout.npz"), **g4)

    # ---------------- G4b: the other patchout variants (maest.py:690-780), train mode, seeded draws -------
    print("G4b")
    g4b = {}
    variants = {
        "tf_u": dict(s_patchout_t=20, s_patchout_f=2, u_patchout=25),
        "interleaved": dict(s_patchout_t_interleaved=2, s_patchout_f_interleaved=2),
        "indices": dict(s_patchout_t_indices=(0, 5, 60), s_patchout_f_indices=(1, 8)),
    }
    for name, kw in variants.items():
        mv, sdv = build(rm, "discogs-maest-10s-pw-129e", 625, **kw)
        mv.train()
        xv = randn((2, 1, 96, 625), 77)
        torch.manual_seed(4242)
        with torch.no_grad():
            lv, fv = mv(xv.clone())
        g4b[f"logits_{name}"] = lv.numpy()
        g4b[f"features_{name}"] = fv.numpy()
        # eval mode too (the fixed-index / interleaved variants apply in eval as well)
        mv.eval()
        with torch.no_grad():
            le, _ = mv(xv.clone())
        g4b[f"logits_eval_{name}"] = le.numpy()
    np.savez(os.path.join(OUT, "g4b_patchout_variants.npz"), **g4b)

    # ---------------- G5: training step (loss + gradients) ---------------------------------
    print("G5")
    B, T, C = 4, 625, 400
    mt, sdt = build(rm, "passt_s_swa_p16_128_ap476", 625, input_t=625, s_patchout_t=30)
    mt.train()
    xt = randn((B, 1, 96, T), 21)
    rng = np.random.Generator(np.random.PCG64(22))
    y = torch.from_numpy((rng.random((B, C)) < 0.02).astype(np.float32))
    perm = torch.from_numpy(rng.permutation(B).astype(np.int64))
    lam_raw = rng.beta(0.3, 0.3, B).astype(np.float32)
    lam = torch.from_numpy(np.maximum(lam_raw, 1 - lam_raw))
    Tp = 61
    torch.manual_seed(555)
    toff = torch.randint(1 + 62 - Tp, (1,)).item()
    keep = torch.randperm(Tp)[: Tp - 30].sort().values
    # reference: module.py:77-90 driven directly on the imported MAEST
    torch.manual_seed(555)
    xm = xt * lam.reshape(B, 1, 1, 1) + xt[perm] * (1.0 - lam.reshape(B, 1, 1, 1))
    ym = y * lam.reshape(B, 1) + y[perm] * (1.0 - lam.reshape(B, 1))
    y_hat, _ = mt(xm)
    y_hat.retain_grad()
    loss = F.binary_cross_entropy_with_logits(y_hat, ym)
    loss.backward()
    ref_grads = {k: p.grad.detach().clone() for k, p in mt.named_parameters() if p.grad is not None}
    # oracle
    sdo = {k: v.clone().requires_grad_(True) for k, v in sdt.items()}
    oloss, ologits = O.training_loss(xt, y, sdo, perm, lam, toffset=toff, t_keep=keep.tolist())
    ologits.retain_grad()
    oloss.backward()
    check("G5 loss", oloss.detach(), loss.detach())
    check("G5 dlogits", ologits.grad, y_hat.grad)
    names = [n for n, _ in O.state_dict_spec(625, 400)]
    gnorm, gprobe, has = [], [], []
    for n in names:
        if n in ref_grads:
            g = ref_grads[n]
            check(f"G5 grad {n}", sdo[n].grad, g, tol=1e-6 * max(1.0, g.abs().max().item()) + 1e-9)
            gnorm.append(g.norm().item())
            gprobe.append(g.flatten()[:8].numpy())
            has.append(1)
        else:
            gnorm.append(0.0)
            gprobe.append(np.zeros(8, np.float32))
            has.append(0)
    np.savez(os.path.join(OUT, "g5_train_step.npz"), loss=loss.detach().numpy(),
             logits=y_hat.detach().numpy(), dlogits=y_hat.grad.numpy(), perm=perm.numpy(),
             lam=lam.numpy(), toffset=np.int64(toff), t_keep=keep.numpy(), y=y.numpy(),
             grad_norm=np.array(gnorm, np.float32), grad_probe=np.stack(gprobe).astype(np.float32),
             grad_present=np.array(has, np.int8),
             grad_qkv0=ref_grads["blocks.0.attn.qkv.weight"][:16, :16].numpy(),
             grad_patch=ref_grads["patch_embed.proj.weight"].reshape(768, 256)[:8].numpy(),
             grad_tpe=ref_grads["time_new_pos_embed"].reshape(768, 62)[:4].numpy())

    # teacher-student ("separated") variant, module.py:280-301
    mts, sdts = build(rm, "discogs-maest-30s-pw-73e-ts", 625, input_t=625, s_patchout_t=30,
                      n_classes=519, distilled_type="separated")
    mts.train()
    C2 = 519
    y2 = torch.from_numpy((rng.random((B, C2)) < 0.02).astype(np.float32))
    yt2 = torch.from_numpy((rng.random((B, C2)) < 0.03).astype(np.float32))
    torch.manual_seed(556)
    toff2 = torch.randint(1 + 62 - Tp, (1,)).item()
    keep2 = torch.randperm(Tp)[: Tp - 30].sort().values
    torch.manual_seed(556)
    xm = xt * lam.reshape(B, 1, 1, 1) + xt[perm] * (1.0 - lam.reshape(B, 1, 1, 1))
    ym = y2 * lam.reshape(B, 1) + y2[perm] * (1.0 - lam.reshape(B, 1))
    ytm = yt2 * lam.reshape(B, 1) + yt2[perm] * (1.0 - lam.reshape(B, 1))
    yh, yht, _ = mts(xm)
    loss_ts = (F.binary_cross_entropy_with_logits(yh, ym)
               + F.binary_cross_entropy_with_logits(yht, ytm)) / 2
    loss_ts.backward()
    sdo = {k: v.clone().requires_grad_(True) for k, v in sdts.items()}
    ol, olc_, old_ = O.training_loss(xt, y2, sdo, perm, lam, toffset=toff2, t_keep=keep2.tolist(),
                                     y_teacher=yt2)
    ol.backward()
    check("G5ts loss", ol.detach(), loss_ts.detach())
    g_hd = mts.head_dist.weight.grad
    check("G5ts grad head_dist", sdo["head_dist.weight"].grad, g_hd, tol=1e-8)
    np.savez(os.path.join(OUT, "g5_train_step_ts.npz"), loss=loss_ts.detach().numpy(),
             logits_cls=yh.detach().numpy(), logits_dist=yht.detach().numpy(), y=y2.numpy(),
             y_teacher=yt2.numpy(), toffset=np.int64(toff2), t_keep=keep2.numpy(),
             perm=perm.numpy(), lam=lam.numpy(),
             grad_head_dist=g_hd[:8, :16].numpy(), grad_head_dist_norm=np.float32(g_hd.norm().item()),
             grad_qkv11_norm=np.float32(mts.blocks[11].attn.qkv.weight.grad.norm().item()),
             grad_patch_norm=np.float32(mts.patch_embed.proj.weight.grad.norm().item()))

    # ---------------- G7: mel restatement (PARITY UNPINNED, see oracle header) -------------
    print("G7")
    rngw = np.random.Generator(np.random.PCG64(31))
    w = torch.from_numpy((rngw.random((2, 160000), dtype=np.float32) * 2 - 1))
    lm = O.logmel(w)
    assert lm.shape == (2, 96, 626)
    w30 = torch.from_numpy((rngw.standard_normal((1, 480000), dtype=np.float32) * 0.1))
    lm30 = O.logmel(w30)
    assert lm30.shape == (1, 96, 1876)
    np.savez(os.path.join(OUT, "g7_mel_restatement_unpinned.npz"),
             logmel_10s_probe=lm[:, :, ::25].numpy(), logmel_10s_mean=lm.mean().numpy(),
             logmel_10s_std=lm.std().numpy(), logmel_30s_probe=lm30[:, :, ::75].numpy(),
             fb_sum=O.mel_filterbank().sum(0), fb_nnz=np.int64((O.mel_filterbank() > 0).sum()))
    print("fb nnz", (O.mel_filterbank() > 0).sum())
    print("done; fixtures in", OUT)
    for f in sorted(os.listdir(OUT)):
        print(f"  {f}: {os.path.getsize(os.path.join(OUT, f))} B")


if __name__ == "__main__":
    main()
