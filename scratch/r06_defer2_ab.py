"""gemm_nt256d_kernel (32x32x16, C tile deferred through an idle ring buffer) against gemm_nt256o_kernel (16x16x32) on the plain bf16-output GEMMs of the
model: persistent form (256 workgroups, no tail launch) and one workgroup per tile; alternating, min of 3 rounds."""
import sys, torch
sys.path.insert(0, ".")
from maest_amd import ops
dev = "cuda"; dt = torch.bfloat16
torch.manual_seed(0)
def mk(r, c, s=1.0): return (torch.randn(r, c, device=dev) * s).to(dt)
def bench(fn, n=10):
    for _ in range(2): fn()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
cases = [("qkv", 2304, 768, True), ("proj", 768, 768, True), ("fc2", 768, 3072, True), ("dqkv", 768, 2304, False), ("fc1 plain", 3072, 768, True)]
for wgs in (256, 0):
    for M in (74240, 143360):
        for nm, N, K, hasb in cases:
            a = mk(M, K); w = mk(N, K, 0.05); bias = torch.randn(N, device=dev) if hasb else None
            out = torch.empty(M, N, device=dev, dtype=dt)
            t = {0: [], 64: []}
            for rnd in range(3):
                for d in (0, 64):
                    with ops.options(gemm_defer=d, gemm_wgs=wgs, gemm_tail=0):
                        t[d].append(bench(lambda: ops.gemm_nt(a, w, bias, out=out)))
            fl = 2.0 * M * N * K / 1e9
            print(f"wgs={wgs:3d} M={M:6d} {nm:9s} N={N:5d} K={K:5d}: 16x16 kernel {min(t[0])*1e3:7.1f} us {fl/min(t[0]):6.1f} TF | deferred {min(t[64])*1e3:7.1f} us {fl/min(t[64]):6.1f} TF | {min(t[0])/min(t[64]):.3f}x", flush=True)
            del a, w, out
