import sys, torch
sys.path.insert(0, ".")
from maest_amd import ops
dev = "cuda"; B, P, Fg, Tt = 256, 288, 12, 62
dx = torch.randn(B * (P + 2), 768, device=dev)
tok = torch.stack([torch.arange(P) // 24, torch.arange(P) % 24], 1).to(torch.int32).to(dev)
def run():
    z = [torch.zeros(n, device=dev) for n in (768, 768, 2 * 768, 768 * Fg, 768 * Tt)]
    return ops.token_assemble_bwd(dx, B, Fg, Tt, 0, tok, torch.bfloat16, z[0], z[1], z[2].view(2, 768), z[3].view(768, Fg), z[4].view(768, Tt))
for _ in range(3): run()
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
import time
ts = []
for _ in range(20):
    z = [torch.zeros(n, device=dev) for n in (768, 768, 2 * 768, 768 * Fg, 768 * Tt)]
    e0.record()
    ops.token_assemble_bwd(dx, B, Fg, Tt, 0, tok, torch.bfloat16, z[0], z[1], z[2].view(2, 768), z[3].view(768, Fg), z[4].view(768, Tt))
    e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
print("token_assemble_bwd", sorted(ts)[len(ts) // 2] * 1e3, "us")
