# How far is gemm_nt256w from what the vendor library reaches on the SAME shapes?  (measurement only: the product
# never calls a library GEMM.)  torch.matmul -> hipBLASLt / rocBLAS bf16, fp32 accumulate, plain bf16 output.
import sys, torch
sys.path.insert(0, ".")
from maest_amd import ops
dev = "cuda"; dt = torch.bfloat16
def bench(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
def mk(r, c): return torch.randn(r, c, device=dev).to(dt)
for M in (256 * 290, 256 * 560, 65536):
    print("M =", M)
    for (nm, N, K) in [("qkv", 2304, 768), ("proj", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072), ("dqkv", 768, 2304), ("big", 4096, 4096)]:
        a = mk(M, K); w = mk(N, K); out = torch.empty(M, N, device=dev, dtype=dt)
        wt = w.t()
        t_lib = bench(lambda: torch.matmul(a, wt, out=out))
        t_own = bench(lambda: ops.gemm_nt(a, w, None, out=out))
        fl = 2.0 * M * N * K
        print(f"  {nm:5s} N={N:5d} K={K:5d}  library {t_lib:7.3f} ms {fl/t_lib/1e9:7.1f} TF/s | gemm_nt {t_own:7.3f} ms {fl/t_own/1e9:7.1f} TF/s")
        del a, w, out
print("wgrad (TN): dW[out][in] = dY^T X, K = tokens")
M = 256 * 290
for (nm, No, Ki) in [("qkv", 2304, 768), ("proj", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072)]:
    dy = mk(M, No); x = mk(M, Ki)
    dw = torch.zeros(No, Ki, device=dev); db = torch.zeros(No, device=dev)
    out = torch.empty(No, Ki, device=dev, dtype=dt)
    dyt = dy.t()
    t_lib = bench(lambda: torch.matmul(dyt, x, out=out))
    t_own = bench(lambda: ops.gemm_tn(dy, x, dw, colsum=db, split_k=0))
    fl = 2.0 * M * No * Ki
    print(f"  {nm:5s} out={No:5d} in={Ki:5d}  library {t_lib:7.3f} ms {fl/t_lib/1e9:7.1f} TF/s | gemm_tn {t_own:7.3f} ms {fl/t_own/1e9:7.1f} TF/s")
