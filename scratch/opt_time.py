import sys, torch, time
sys.path.insert(0, ".")
from maest_amd import get_maest
from maest_amd.module import Module
dev = "cuda"
net = get_maest("passt_s_swa_p16_128_ap476", pretrained=False, input_t=625, s_patchout_t=30, precision="bf16").to(dev).train()
mod = Module(net=net)
B = 256
x = torch.randn(B, 1, 96, 626, device=dev); y = (torch.rand(B, 400, device=dev) < 0.006).float()
for fused in (None, True):
    kw = {} if fused is None else {"fused": True}
    opt = torch.optim.AdamW(mod.parameters(), lr=2e-5, weight_decay=1e-4, **kw)
    def phases():
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        ev[0].record(); loss = mod.training_step((x, None, y), 0); ev[1].record()
        loss.backward(); ev[2].record(); opt.step(); ev[3].record(); opt.zero_grad(); ev[4].record()
        torch.cuda.synchronize()
        return [ev[i].elapsed_time(ev[i+1]) for i in range(4)]
    for _ in range(3): phases()
    import numpy as np
    r = np.array([phases() for _ in range(5)]).mean(0)
    t0 = time.perf_counter()
    for _ in range(5):
        loss = mod.training_step((x, None, y), 0); loss.backward(); opt.step(); opt.zero_grad()
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 5 * 1e3
    print(f"fused={fused}: fwd {r[0]:.2f} bwd {r[1]:.2f} opt.step {r[2]:.2f} zero_grad {r[3]:.2f} ms | wall {wall:.2f} ms/step")
