# where does the C-tile epilogue time go?  gemm_ablate: 0 = real, 1 = no drain (LDS staging only), 2 = drain into a
# 256-row window (same store instructions, L2-resident lines)
import sys, torch
sys.path.insert(0, ".")
from maest_amd import ops
dev = "cuda"; dt = torch.bfloat16
M = 256 * 290
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
def mk(r, c): return torch.randn(r, c, device=dev).to(dt)
for (nm, N, K, epi) in [("qkv", 2304, 768, "none"), ("proj", 768, 768, "res"), ("fc1", 3072, 768, "gelu"), ("dfc1", 768, 3072, "none"), ("dproj", 768, 768, "none")]:
    a = mk(M, K); w = mk(N, K); bias = torch.randn(N, device=dev)
    if epi == "none":
        out = torch.empty(M, N, device=dev, dtype=dt); fn = lambda: ops.gemm_nt(a, w, bias, out=out)
    elif epi == "res":
        out = torch.empty(M, N, device=dev); res = torch.randn(M, N, device=dev)
        fn = lambda: ops.gemm_nt(a, w, bias, out=out, epi=ops.EPI_RESIDUAL, aux_in=res)
    else:
        out = torch.empty(M, N, device=dev, dtype=dt); aux = torch.empty(M, N, device=dev, dtype=dt)
        fn = lambda: ops.gemm_nt(a, w, bias, out=out, epi=ops.EPI_GELU, aux_out=aux)
    line = f"{nm:6s} N={N:4d} K={K:4d} {epi:5s}"
    for v in (0, 1, 2):
        ops.set_option("gemm_ablate", v)
        ms = bench(fn)
        line += f" | ablate={v}: {ms*1e3:6.1f} us {2.0*M*N*K/ms/1e9:6.0f} TF"
    ops.set_option("gemm_ablate", 0)
    print(line, flush=True)
