import sys, torch
sys.path.insert(0, ".")
from maest_amd import ops
dev = "cuda"; dt = torch.bfloat16
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for (M, N) in [(3072, 768), (768, 3072), (2304, 768), (768, 768)]:
    for K in (7168, 14336, 28672, 74240):
        dy = torch.randn(K, M, device=dev).to(dt); x = torch.randn(K, N, device=dev).to(dt)
        dw = torch.zeros(M, N, device=dev); db = torch.zeros(M, device=dev)
        for sk in (0,):
            ms = bench(lambda: ops.gemm_tn(dy, x, dw, colsum=db, split_k=sk))
            ms2 = bench(lambda: ops.gemm_tn(dy, x, dw, colsum=None, split_k=sk))
            print(f"TN M={M} N={N} K={K} sk={sk}: {ms:7.3f} ms {2.0*M*N*K/ms/1e9:7.1f} TF/s | no colsum {ms2:7.3f} ms {2.0*M*N*K/ms2/1e9:7.1f} TF/s")
