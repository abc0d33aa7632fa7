# stress: the B = 256 fp32 training forward (+ backward) repeated; any loss further than 1e-6 from the first is reported
import sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import test_fullsize_gpu as T
from maest_amd.module import Module
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
tail = (sys.argv[2] != "0") if len(sys.argv) > 2 else True
bwd = (sys.argv[3] != "0") if len(sys.argv) > 3 else True
net, _ = T._model("fp32", input_t=625, s_patchout_t=30)
net.train(); net._engine.head_tail = tail
mod = Module(net=net, mixup_alpha=0.3)
B, Tt = 256, 626
x = T.randn((B, 1, 96, Tt), 21).to("cuda")
rng = np.random.Generator(np.random.PCG64(22))
y = torch.from_numpy((rng.random((B, 400)) < 0.00625).astype(np.float32)).to("cuda")
perm = torch.from_numpy(rng.permutation(B)); lam = torch.from_numpy(np.maximum(b := rng.beta(0.3, 0.3, B).astype(np.float32), 1 - b))
Tp = (Tt - 16) // 10 + 1
keep = torch.from_numpy(np.sort(rng.permutation(Tp)[: Tp - 30]))
ref, bad = None, 0
for it in range(n):
    for p in net.parameters(): p.grad = None
    loss = mod.training_step((x, None, y), 0, _mixup=(perm, lam), _patchout=(0, keep))
    if bwd: loss.backward()
    l = loss.item()
    ref = l if ref is None else ref
    if abs(l - ref) > 1e-6:
        bad += 1; print("deviant at", it, repr(l), "vs", repr(ref), flush=True)
print(f"head_tail={tail} backward={bwd}: {bad} deviants in {n} steps (reference {ref!r})")
