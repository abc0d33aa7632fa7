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
M, N, K = 3072, 768, 74240
dy = torch.randn(K, M, device=dev).to(dt); x = torch.randn(K, N, device=dev).to(dt)
dw = torch.zeros(M, N, device=dev)
for sk in (1, 2, 3, 4, 5, 6, 7, 8, 10, 14):
    ms = bench(lambda: ops.gemm_tn(dy, x, dw, colsum=None, split_k=sk))
    slices = K / 32 / sk
    print(f"sk={sk:2d} WGs={36*sk:4d}: {ms:7.3f} ms  {2.0*M*N*K/ms/1e9:7.1f} TF/s   {ms*1e3/slices:6.3f} us/slice")
