import sys, os, torch
sys.path.insert(0, ".")
from maest_amd import ops
dev = "cuda"; dt = torch.bfloat16
def bench(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
def mk(r, c, dtype=dt): return torch.randn(r, c, device=dev).to(dtype)
for M in (256 * 290, 256 * 560):
    print("M =", M)
    for (nm, N, K) in [("qkv", 2304, 768), ("proj", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072), ("dqkv", 768, 2304)]:
        a = mk(M, K); w = mk(N, K); bias = torch.randn(N, device=dev)
        out_bf = torch.empty(M, N, device=dev, dtype=dt); out32 = torch.empty(M, N, device=dev)
        res = torch.randn(M, N, device=dev); aux = torch.empty(M, N, device=dev, dtype=dt)
        cases = [("none->bf16", lambda: ops.gemm_nt(a, w, bias, out=out_bf)),
                 ("resid->f32", lambda: ops.gemm_nt(a, w, bias, out=out32, epi=ops.EPI_RESIDUAL, aux_in=res)),
                 ("gelu->bf16", lambda: ops.gemm_nt(a, w, bias, out=out_bf, epi=ops.EPI_GELU)),
                 ("gelu+aux", lambda: ops.gemm_nt(a, w, bias, out=out_bf, epi=ops.EPI_GELU, aux_out=aux)),
                 ("mul->bf16", lambda: ops.gemm_nt(a, w, None, out=out_bf, epi=ops.EPI_MUL, aux_in=aux))]
        for cn, fn in cases:
            r = []
            for v in ("0", "1", "2"):
                ops.set_option("gemm_epilogue", int(v))
                ms = bench(fn); r.append(ms)
            fl = 2.0 * M * N * K
            print(f"  {nm:5s} {cn:11s} 2pass: {r[0]:7.3f} ms | 4pass-dbuf: {r[1]:7.3f} ms | 1pass: {r[2]:7.3f} ms   best {fl/min(r)/1e9:7.1f} TF/s")
        del a, w, out_bf, out32, res, aux
