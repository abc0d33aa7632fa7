"""Column-panel tile order of gemm_nt256o_kernel (MAEST_GEMM_PANEL): stand-alone GEMMs at the model's wide shapes, panel 0 (row-major)
vs automatic vs forced widths, persistent form as the engine launches it (gemm_wgs = 256, no tail launch), alternating, min of 3 rounds."""
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
cases = [("qkv  none", 2304, 768, {}), ("fc1  gelu+aux", 3072, 768, dict(pair=True)), ("fc1  gelu", 3072, 768, dict(gelu=True)),
         ("dfc2 mul", 3072, 768, dict(mul=True)), ("fc2  none", 768, 3072, {}), ("proj none", 768, 768, {})]
for M in (74240, 143360):
    for nm, N, K, o in cases:
        a = mk(M, K); w = mk(N, K, 0.05); bias = torch.randn(N, device=dev)
        out = torch.empty(M, N, device=dev, dtype=dt)
        kw = dict(out=out)
        if o.get("pair"): kw.update(epi=ops.EPI_GELU, aux_out=torch.empty(M, N, device=dev, dtype=dt))
        if o.get("gelu"): kw.update(epi=ops.EPI_GELU)
        if o.get("mul"): kw.update(epi=ops.EPI_MUL, aux_in=mk(M, N))
        widths = (0, -1) + ((3, 4, 6) if N == 3072 else ((3, 5) if N == 2304 else ()))
        t = {w_: [] for w_ in widths}
        for rnd in range(3):
            for w_ in widths:
                with ops.options(gemm_panel=w_, gemm_wgs=256, gemm_tail=0):
                    t[w_].append(bench(lambda: ops.gemm_nt(a, w, bias, **kw)))
        fl = 2.0 * M * N * K / 1e9
        print(f"M={M:6d} {nm:14s} N={N:5d} K={K:5d}: " + " | ".join(f"panel {w_:2d}: {min(v)*1e3:7.1f} us {fl/min(v):6.1f} TF" for w_, v in t.items()), flush=True)
        del a, w, out, kw
