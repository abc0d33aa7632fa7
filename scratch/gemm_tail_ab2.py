"""last partial round of the 256-row-tile NT GEMM: 128-row tail tiles (gemm_tail = 1, the 8-wave MTW = 2 kernel) against none (0) with the
one-wave-per-SIMD kernel on the full tiles, N = 768 shapes at M = 74240 (870 tiles = 3.4 rounds)"""
import sys, torch
sys.path.insert(0, ".")
from maest_amd import ops
dev = "cuda"; dt = torch.bfloat16
def bench(fn, n=10):
    for _ in range(2): fn()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for M in (74240, 143360):
  for nm, N, K, res in (("proj", 768, 768, True), ("dprj", 768, 768, False), ("dqkv", 768, 2304, False), ("fc2", 768, 3072, True), ("dfc1", 768, 3072, False),
                      ("qkv", 2304, 768, False), ("fc1", 3072, 768, False)):
    a = torch.randn(M, K, device=dev).to(dt); w = (torch.randn(N, K, device=dev) * 0.05).to(dt); bias = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev, dtype=torch.float32 if res else dt)
    kw = dict(out=out)
    if res: kw.update(epi=ops.EPI_RESIDUAL, aux_in=torch.randn(M, N, device=dev))
    t = {0: [], 1: []}
    for rnd in range(3):
        for v in (0, 1):
            with ops.options(gemm_tail=v):
                t[v].append(bench(lambda: ops.gemm_nt(a, w, bias, **kw)))
    print(f"M={M} {nm:5s} N={N} K={K}: tail tiles {min(t[1])*1e3:7.1f} us | none {min(t[0])*1e3:7.1f} us", flush=True)
    del a, w, out, kw
