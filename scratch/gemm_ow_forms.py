"""gemm_nt256o_kernel vs the 8-wave kernel, per epilogue form at the training step's shapes (M = 74240), paired, min of 3 rounds."""
import sys, torch
sys.path.insert(0, ".")
from maest_amd import ops
dev = "cuda"; dt = torch.bfloat16
torch.manual_seed(0)
def mk(r, c, dtype=dt, s=1.0): return (torch.randn(r, c, device=dev) * s).to(dtype)
def bench(fn, n=10):
    for _ in range(2): fn()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
M = 74240
cases = [("qkv  none->bf16", 2304, 768, dict()), ("proj resid->f32", 768, 768, dict(res=True)), ("fc1  gelu+aux  ", 3072, 768, dict(pair=True)),
         ("fc1  gelu      ", 3072, 768, dict(gelu=True)), ("fc2  resid->f32", 768, 3072, dict(res=True)), ("dfc2 mul->bf16 ", 3072, 768, dict(mul=True)),
         ("dfc1 none->bf16", 768, 3072, dict()), ("dfc1 none->f32 ", 768, 3072, dict(f32=True)), ("dqkv none->bf16", 768, 2304, dict()),
         ("dqkv resid->f32", 768, 2304, dict(res=True)), ("dprj none->bf16", 768, 768, dict())]
for nm, N, K, o in cases:
    a = mk(M, K); w = mk(N, K, s=0.05); bias = torch.randn(N, device=dev)
    odt = torch.float32 if (o.get("res") or o.get("f32")) else dt
    out = torch.empty(M, N, device=dev, dtype=odt)
    kw = dict(out=out)
    if o.get("res"): kw.update(epi=ops.EPI_RESIDUAL, aux_in=torch.randn(M, N, device=dev))
    if o.get("pair"): kw.update(epi=ops.EPI_GELU, aux_out=torch.empty(M, N, device=dev, dtype=dt))
    if o.get("gelu"): kw.update(epi=ops.EPI_GELU)
    if o.get("mul"): kw.update(epi=ops.EPI_MUL, aux_in=mk(M, N))
    t = {0: [], 3: []}
    for rnd in range(3):
        for v in (0, 3):
            with ops.options(gemm_variant=v):
                t[v].append(bench(lambda: ops.gemm_nt(a, w, bias, **kw)))
    fl = 2.0 * M * N * K / 1e9
    print(f"{nm} N={N:5d} K={K:5d}: one-wave {min(t[0])*1e3:7.1f} us {fl/min(t[0]):7.1f} TF/s | 8-wave {min(t[3])*1e3:7.1f} us {fl/min(t[3]):7.1f} TF/s | {min(t[3])/min(t[0]):.3f}x", flush=True)
    del a, w, out, kw
