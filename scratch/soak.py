# 300 training steps at the bench shape (bf16, batch 256, fresh mixup / patchout draws every step): loss must stay
# finite and fall; weights must stay finite.
import sys, torch, numpy as np
sys.path.insert(0, ".")
from maest_amd import get_maest
from maest_amd.module import Module
dev = "cuda"
torch.manual_seed(0); np.random.seed(0)
net = get_maest("passt_s_swa_p16_128_ap476", pretrained=False, input_t=625, s_patchout_t=30, precision="bf16").to(dev).train()
mod = Module(net=net, lr=1e-4); opt = mod.get_optimizer()
B = 256
x = torch.randn(B, 1, 96, 626, device=dev); y = (torch.rand(B, 400, device=dev) < 0.00625).float()
losses = []
for it in range(300):
    loss = mod.training_step((x, None, y), it); loss.backward(); opt.step(); opt.zero_grad()
    if it % 25 == 0 or it == 299:
        losses.append(loss.item()); print(it, round(losses[-1], 5), flush=True)
assert all(np.isfinite(losses)) and losses[-1] < 0.2 * losses[0]
assert all(torch.isfinite(p).all() for p in net.parameters())
print("soak ok")
