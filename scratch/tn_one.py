import sys, torch
sys.path.insert(0, ".")
from maest_amd import ops
dev = "cuda"; dt = torch.bfloat16
M = 256 * 290
dy = torch.randn(M, 3072, device=dev).to(dt); x = torch.randn(M, 768, device=dev).to(dt)
dw = torch.zeros(3072, 768, device=dev); db = torch.zeros(3072, device=dev)
a = torch.randn(M, 768, device=dev).to(dt); w = torch.randn(3072, 768, device=dev).to(dt)
out = torch.empty(M, 3072, device=dev, dtype=dt)
for _ in range(3):
    ops.gemm_tn(dy, x, dw, colsum=db, split_k=0)
    ops.gemm_nt(a, w, None, out=out)
torch.cuda.synchronize()
