import sys, torch
sys.path.insert(0, ".")
from maest_amd import ops
dev = "cuda"; dt = torch.bfloat16
M = 256 * 290
which = sys.argv[1] if len(sys.argv) > 1 else "fc2"
N, K = {"fc2": (768, 3072), "fc1": (3072, 768), "qkv": (2304, 768)}[which]
a = torch.randn(M, K, device=dev).to(dt); w = torch.randn(N, K, device=dev).to(dt); bias = torch.randn(N, device=dev)
out = torch.empty(M, N, device=dev, dtype=dt)
for _ in range(5): ops.gemm_nt(a, w, bias, out=out)
dy = torch.randn(M, N, device=dev).to(dt); dw = torch.zeros(N, K, device=dev); db = torch.zeros(N, device=dev)
for _ in range(5): ops.gemm_tn(dy, a, dw, colsum=db, split_k=7)
torch.cuda.synchronize()
