import sys, torch
sys.path.insert(0, ".")
from maest_amd import ops
dev="cuda"; dt=torch.bfloat16
def bench(name, fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n
M=74240
for N in (768, 2304, 3072):
    for K in (64, 256, 768, 1536, 3072):
        a=torch.randn(M,K,device=dev).to(dt); b=torch.randn(N,K,device=dev).to(dt)
        out=torch.empty(M,N,device=dev,dtype=dt); out32=torch.empty(M,N,device=dev)
        t1=bench("", lambda: ops.gemm_nt(a,b,None,out=out))
        t2=bench("", lambda: ops.gemm_nt(a,b,None,out=out32))
        print(f"N={N:5d} K={K:5d}  bf16-out {t1*1e3:8.1f} us  fp32-out {t2*1e3:8.1f} us   ({2.0*M*N*K/t1/1e9:7.1f} TF/s)")
