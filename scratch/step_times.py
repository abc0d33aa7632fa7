import sys, time, torch
sys.path.insert(0, ".")
from maest_amd import get_maest
from maest_amd.module import Module
dev = "cuda"
net = get_maest("passt_s_swa_p16_128_ap476", pretrained=False, input_t=625, s_patchout_t=30, precision="bf16").to(dev).train()
mod = Module(net=net); opt = mod.configure_optimizers()
x = torch.randn(256, 1, 96, 626, device=dev); y = (torch.rand(256, 400, device=dev) < 0.006).float()
ts = []
for i in range(30):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    loss = mod.training_step((x, None, y), 0); loss.backward(); opt.step(); opt.zero_grad()
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print(" ".join(f"{t:.1f}" for t in ts))
import subprocess
print(subprocess.run("rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|Power' | head -4", shell=True, capture_output=True, text=True).stdout)
