# the hot kernels of one training step at the bench shape (B = 256, N = 290), each launched a few times on real
# (random) data: the workload of the PMC passes (scratch/pmc_util.sh -> profiles/r02_mfma_util.json)
import sys, torch
sys.path.insert(0, ".")
from maest_amd import ops
dev = "cuda"; dt = torch.bfloat16
B, N = 256, 290
M = B * N
def mk(r, c): return torch.randn(r, c, device=dev).to(dt)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for (nm, Nn, K, epi) in [("qkv", 2304, 768, "none"), ("proj", 768, 768, "res"), ("fc1", 3072, 768, "gelu"), ("fc2", 768, 3072, "res"),
                         ("dfc2", 3072, 768, "mul"), ("dfc1", 768, 3072, "none"), ("dproj", 768, 768, "none"), ("dqkv", 768, 2304, "none")]:
    a = mk(M, K); w = mk(Nn, K); bias = torch.randn(Nn, device=dev)
    for _ in range(reps):
        if epi == "none":
            ops.gemm_nt(a, w, bias, out_dtype=dt)
        elif epi == "res":
            res = torch.randn(M, Nn, device=dev)
            ops.gemm_nt(a, w, bias, out_dtype=torch.float32, epi=ops.EPI_RESIDUAL, aux_in=res)
        elif epi == "gelu":
            aux = torch.empty(M, Nn, device=dev, dtype=dt)
            ops.gemm_nt(a, w, bias, out_dtype=dt, epi=ops.EPI_GELU, aux_out=aux)
        else:
            aux = mk(M, Nn)
            ops.gemm_nt(a, w, None, out_dtype=dt, epi=ops.EPI_MUL, aux_in=aux)
    del a, w
for (nm, Nn, K) in [("qkv", 2304, 768), ("proj", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072)]:
    dy = mk(M, Nn); xx = mk(M, K)
    dw = torch.zeros(Nn, K, device=dev); db = torch.zeros(Nn, device=dev)
    for _ in range(reps):
        ops.gemm_tn(dy, xx, dw, colsum=db, split_k=0)
    del dy, xx
qkv = mk(M, 2304)
for _ in range(reps):
    out, lse = ops.attn_fwd(qkv, B, N, 0.125, save_lse=True)
do = torch.randn_like(out)
for _ in range(reps):
    ops.attn_bwd(qkv, out, do, lse, B, N, 0.125)
x = torch.randn(M, 768, device=dev); g = torch.ones(768, device=dev); b = torch.zeros(768, device=dev)
dg = torch.zeros(768, device=dev); db = torch.zeros(768, device=dev); dres = torch.randn(M, 768, device=dev); dy = mk(M, 768)
for _ in range(reps):
    y, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-6, dt, save_stats=True)
    ops.layernorm_bwd(dy, x, g, mean, rstd, dres, dg, db, lp_dtype=dt)
torch.cuda.synchronize()
print("done")
