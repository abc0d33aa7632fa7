"""gemm_nt256o_kernel with v_mfma_f32_16x16x32_bf16 (the product library) against its 32x32x16 form (scratch/pw_abl/libmaest_o-mfma32.so, built from
commit ddaed71's kernel), every epilogue form the model uses, persistent launch; alternating, min of 3 rounds."""
import sys, ctypes, torch
sys.path.insert(0, ".")
from maest_amd import ops, _lib
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
new = _lib.load()
old = _lib._bind(ctypes.CDLL("scratch/pw_abl/libmaest_o-mfma32.so"))
cases = [("qkv  none", 2304, 768, {}), ("proj none", 768, 768, {}), ("fc2  none", 768, 3072, {}), ("dqkv none", 768, 2304, {}),
         ("fc1  gelu", 3072, 768, dict(gelu=True)), ("fc1  gelu+aux", 3072, 768, dict(pair=True)), ("dfc2 mul", 3072, 768, dict(mul=True)),
         ("dprj rowdot", 768, 768, dict(rowdot=True))]
tot = {"old": 0.0, "new": 0.0}
for M, ntok in ((74240, 290), (143360, 560)):
    for nm, N, K, o in cases:
        a = mk(M, K); w = mk(N, K, 0.05); bias = torch.randn(N, device=dev)
        out = torch.empty(M, N, device=dev, dtype=dt)
        kw = dict(out=out)
        if o.get("pair"): kw.update(epi=ops.EPI_GELU, aux_out=torch.empty(M, N, device=dev, dtype=dt))
        if o.get("gelu"): kw.update(epi=ops.EPI_GELU)
        if o.get("mul"): kw.update(epi=ops.EPI_MUL, aux_in=mk(M, N))
        other = mk(M, N) if o.get("rowdot") else None
        def call():
            if other is not None: return ops.gemm_nt_rowdot(a, w, other, ntok, out_dtype=dt, bias=bias)
            return ops.gemm_nt(a, w, bias, **kw)
        t = {"old": [], "new": []}
        for rnd in range(3):
            for name, lib in (("old", old), ("new", new)):
                _lib._lib = lib; ops._option_cache.clear()
                with ops.options(gemm_wgs=256, gemm_tail=0):
                    t[name].append(bench(call))
        _lib._lib = new; ops._option_cache.clear()
        fl = 2.0 * M * N * K / 1e9
        tot["old"] += min(t["old"]); tot["new"] += min(t["new"])
        print(f"M={M:6d} {nm:14s} N={N:5d} K={K:5d}: 32x32x16 {min(t['old'])*1e3:7.1f} us {fl/min(t['old']):6.1f} TF | 16x16x32 {min(t['new'])*1e3:7.1f} us {fl/min(t['new']):6.1f} TF | {min(t['old'])/min(t['new']):.3f}x", flush=True)
        del a, w, out, kw
print(f"sum: 32x32x16 {tot['old']*1e3:.1f} us, 16x16x32 {tot['new']*1e3:.1f} us, {tot['old']/tot['new']:.3f}x")
