import sys, torch
sys.path.insert(0, ".")
from maest_amd import ops
dev="cuda"; dt=torch.bfloat16
def bench(name, fn, flops, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/n
    print(f"{name:34s} {ms:8.3f} ms {flops/ms/1e9:8.1f} TF/s")
for S in (4096, 8192):
    a=torch.randn(S,S,device=dev).to(dt); b=torch.randn(S,S,device=dev).to(dt)
    out=torch.empty(S,S,device=dev,dtype=dt)
    bench(f"square {S} randn", lambda: ops.gemm_nt(a,b,None,out=out), 2.0*S**3)
    a.zero_(); b.zero_()
    bench(f"square {S} zeros", lambda: ops.gemm_nt(a,b,None,out=out), 2.0*S**3)
# model shapes with M multiple of 256*256 CUs granularity
for (M,N,K) in [(65536,768,3072),(65536,3072,768),(74240,768,3072)]:
    a=torch.randn(M,K,device=dev).to(dt); b=torch.randn(N,K,device=dev).to(dt); out=torch.empty(M,N,device=dev,dtype=dt)
    bench(f"M={M} N={N} K={K}", lambda: ops.gemm_nt(a,b,None,out=out), 2.0*M*N*K)
