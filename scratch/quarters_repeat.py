# repeat the B = 256 "mean of its quarters" check and print the losses (hunting a rare deviation seen once in the suite)
import sys, numpy as np, torch
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import test_fullsize_gpu as T
from maest_amd.module import Module
net, _ = T._model("fp32", input_t=625, s_patchout_t=30)
net.train()
mod = Module(net=net, mixup_alpha=0.3)
B, Tt = 256, 626
x = T.randn((B, 1, 96, Tt), 21).to("cuda")
rng = np.random.Generator(np.random.PCG64(22))
y = torch.from_numpy((rng.random((B, 400)) < 0.00625).astype(np.float32)).to("cuda")
Q = B // 4
perm = torch.cat([torch.from_numpy(rng.permutation(Q)) + q * Q for q in range(4)])
lam = torch.from_numpy(np.maximum(b := rng.beta(0.3, 0.3, B).astype(np.float32), 1 - b))
Tp = (Tt - 16) // 10 + 1
keep = torch.from_numpy(np.sort(rng.permutation(Tp)[: Tp - 30]))
po = (0, keep)        # full-width input: the only offset that leaves 62 columns of the 62-column table
def step(xs, ys, mix):
    for p in net.parameters(): p.grad = None
    loss = mod.training_step((xs, None, ys), 0, _mixup=mix, _patchout=po)
    loss.backward()
    return loss.item()
seen = set()
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
    lf = step(x, y, (perm, lam))
    lq = [step(x[q*Q:(q+1)*Q], y[q*Q:(q+1)*Q], (perm[q*Q:(q+1)*Q] - q*Q, lam[q*Q:(q+1)*Q])) for q in range(4)]
    key = (lf, tuple(lq))
    if key not in seen:
        seen.add(key); print(rep, "full", repr(lf), "quarters", [repr(v) for v in lq], "mean", sum(lq) / 4, flush=True)
print("distinct outcomes:", len(seen))
