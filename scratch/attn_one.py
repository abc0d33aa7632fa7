import sys, torch
sys.path.insert(0, ".")
from maest_amd import ops
B, N = 256, 560
qkv = torch.randn(B * N, 2304, device="cuda").to(torch.bfloat16)
for mode in (2, 3):
    with ops.options(attn_fwd=mode):
        for _ in range(3): ops.attn_fwd(qkv, B, N, 0.125)
torch.cuda.synchronize()
