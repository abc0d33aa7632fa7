"""One GEMM form, a few launches (for rocprofv3 --pmc passes): python scratch/r06_fc1_one.py <N> <K> <form> [M]"""
import sys, torch
sys.path.insert(0, ".")
from maest_amd import ops
N, K, form = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
M = int(sys.argv[4]) if len(sys.argv) > 4 else 74240
dev = "cuda"; dt = torch.bfloat16
torch.manual_seed(0)
a = torch.randn(M, K, device=dev).to(dt); w = (torch.randn(N, K, device=dev) * 0.05).to(dt); bias = torch.randn(N, device=dev)
out = torch.empty(M, N, device=dev, dtype=dt)
kw = dict(out=out)
if form == "pair": kw.update(epi=ops.EPI_GELU, aux_out=torch.empty(M, N, device=dev, dtype=dt))
if form == "mul": kw.update(epi=ops.EPI_MUL, aux_in=torch.randn(M, N, device=dev).to(dt))
with ops.options(gemm_wgs=256, gemm_tail=0):
    for _ in range(6):
        ops.gemm_nt(a, w, bias, **kw)
torch.cuda.synchronize()
