"""persistent gemm_nt256o_kernel (libmaest_pers) against the one-tile-per-workgroup form (libmaest_np), per epilogue form, M = 74240"""
import sys, glob, ctypes, torch
sys.path.insert(0, ".")
from maest_amd import ops, _lib
dev = "cuda"; dt = torch.bfloat16
torch.manual_seed(0)
libs = {n: _lib._bind(ctypes.CDLL(f"scratch/pw_abl/libmaest_{n}.so")) for n in ("np", "pers")}
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
         ("dfc1 none->bf16", 768, 3072, dict()), ("dqkv none->bf16", 768, 2304, dict()), ("dprj none->bf16", 768, 768, dict())]
for nm, N, K, o in cases:
    a = mk(M, K); w = mk(N, K, s=0.05); bias = torch.randn(N, device=dev)
    odt = torch.float32 if o.get("res") else dt
    out = {n: torch.empty(M, N, device=dev, dtype=odt) for n in libs}
    kw = {}
    if o.get("res"): kw.update(epi=ops.EPI_RESIDUAL, aux_in=torch.randn(M, N, device=dev))
    auxo = {n: None for n in libs}
    if o.get("pair"):
        auxo = {n: torch.empty(M, N, device=dev, dtype=dt) for n in libs}; kw.update(epi=ops.EPI_GELU)
    if o.get("gelu"): kw.update(epi=ops.EPI_GELU)
    if o.get("mul"): kw.update(epi=ops.EPI_MUL, aux_in=mk(M, N))
    t = {n: [] for n in libs}
    for rnd in range(3):
        for n, lib in libs.items():
            _lib._lib = lib
            k2 = dict(kw, out=out[n]);
            if auxo[n] is not None: k2["aux_out"] = auxo[n]
            t[n].append(bench(lambda: ops.gemm_nt(a, w, bias, **k2)))
    eq = torch.equal(out["np"], out["pers"]) and (auxo["np"] is None or torch.equal(auxo["np"], auxo["pers"]))
    print(f"{nm} N={N:5d} K={K:5d}: persistent {min(t['pers'])*1e3:7.1f} us | one tile per workgroup {min(t['np'])*1e3:7.1f} us | {min(t['np'])/min(t['pers']):.3f}x  bit-equal {eq}", flush=True)
    del a, w, out, kw
