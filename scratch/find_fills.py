# who launches the ~39 small fill kernels per training step?  torch.profiler with stacks, aten::zero_/fill_/zeros
import sys, torch
sys.path.insert(0, ".")
from maest_amd import get_maest
from maest_amd.module import Module
from torch.profiler import profile, ProfilerActivity
dev = "cuda"
net = get_maest("passt_s_swa_p16_128_ap476", pretrained=False, input_t=625, s_patchout_t=30, precision="bf16").to(dev).train()
mod = Module(net=net); opt = mod.get_optimizer()
B = 32
x = torch.randn(B, 1, 96, 626, device=dev); y = (torch.rand(B, 400, device=dev) < 0.006).float()
def step():
    loss = mod.training_step((x, None, y), 0); loss.backward(); opt.step(); opt.zero_grad()
for _ in range(3): step()
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    step()
import collections
c = collections.Counter()
for e in prof.events():
    if e.name in ("aten::zero_", "aten::fill_", "aten::zeros", "aten::zeros_like"):
        st = [s for s in e.stack if "maest_amd" in s or "bench" in s or "optim" in s or "autograd" in s][:3]
        c[(e.name, tuple(st))] += 1
for k, v in c.most_common(12):
    print(v, k)
