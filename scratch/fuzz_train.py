# odd-shape fuzz of the training step (fp32 parity mode) against the oracle: loss and a few gradients
import sys, numpy as np, torch
sys.path.insert(0, ".")
from maest_amd import get_maest
from maest_amd.module import Module
from oracle import maest_oracle as O
dev = "cuda"
rng = np.random.Generator(np.random.PCG64(1))
worst = 0
for B, T, po in [(1, 100, 3), (3, 333, 10), (2, 46, 1), (5, 626, 30), (1, 626, 55)]:
    sd = O.make_state_dict(625, seed=B * 1000 + T)
    net = get_maest("passt_s_swa_p16_128_ap476", pretrained=False, input_t=625, s_patchout_t=po, precision="fp32")
    net.load_state_dict(sd); net = net.to(dev).train()
    mod = Module(net=net, mixup_alpha=0.3)
    x = torch.from_numpy(rng.standard_normal((B, 1, 96, T), dtype=np.float32))
    y = torch.from_numpy((rng.random((B, 400)) < 0.02).astype(np.float32))
    perm = torch.from_numpy(rng.permutation(B)); lam = torch.from_numpy(rng.uniform(0.5, 1, B).astype(np.float32))
    Tp = (T - 16) // 10 + 1
    keep = sorted(rng.permutation(Tp)[: Tp - po].tolist())
    toff = int(rng.integers(0, 62 - Tp + 1))
    loss = mod.training_step((x.to(dev), None, y.to(dev)), 0, _mixup=(perm, lam), _patchout=(toff, torch.tensor(keep)))
    loss.backward()
    sdo = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    want, _ = O.training_loss(x, y, sdo, perm, lam, toffset=toff, t_keep=keep)
    want.backward()
    e = abs(loss.item() - want.item()) / abs(want.item())
    for n, p in net.named_parameters():
        if p.grad is None or sdo[n].grad is None: continue
        r = sdo[n].grad
        ge = ((p.grad.cpu() - r).abs().max() / r.abs().max().clamp_min(1e-12)).item()
        e = max(e, ge)
    worst = max(worst, e)
    print(f"B={B} T={T} patchout={po} tokens={2 + 9 * (Tp - po)} toffset={toff}: worst rel err (loss, all grads) {e:.2e}")
print("worst", worst); assert worst < 2e-3
