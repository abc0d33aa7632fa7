# C-tile store cache policy (MAEST_OPT_GEMM_STORE: 0 nt, 1 sc1, 2 plain) x every ViT GEMM shape
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
vals = [0, 1, 2]
tot = {v: 0.0 for v in vals}
for (nm, N, K, epi) in [("qkv", 2304, 768, "none"), ("proj", 768, 768, "none"), ("fc1", 3072, 768, "gelu"), ("fc2", 768, 3072, "none"),
                        ("dfc2", 3072, 768, "mul"), ("dfc1", 768, 3072, "none"), ("dproj", 768, 768, "none"), ("dqkv", 768, 2304, "none"),
                        ("proj-res", 768, 768, "res")]:
    a = mk(M, K); w = mk(N, K); bias = torch.randn(N, device=dev)
    if epi == "none":
        out = torch.empty(M, N, device=dev, dtype=dt); fn = lambda: ops.gemm_nt(a, w, bias, out=out)
    elif epi == "res":
        out = torch.empty(M, N, device=dev); res = torch.randn(M, N, device=dev)
        fn = lambda: ops.gemm_nt(a, w, bias, out=out, epi=ops.EPI_RESIDUAL, aux_in=res)
    elif epi == "gelu":
        out = torch.empty(M, N, device=dev, dtype=dt); aux = torch.empty(M, N, device=dev, dtype=dt)
        fn = lambda: ops.gemm_nt(a, w, bias, out=out, epi=ops.EPI_GELU, aux_out=aux)
    else:
        out = torch.empty(M, N, device=dev, dtype=dt); aux = mk(M, N)
        fn = lambda: ops.gemm_nt(a, w, None, out=out, epi=ops.EPI_MUL, aux_in=aux)
    ref = None
    line = f"{nm:8s} N={N:4d} K={K:4d} {epi:5s}"
    for rep in range(2):
        for v in vals:
            ops.set_option("gemm_store", v)
            ms = bench(fn)
            if rep == 1:
                if ref is None: ref = out.clone()
                else: assert torch.equal(ref, out), "store policy changed the result"
                tot[v] += ms
                line += f" | policy {v}: {ms*1e3:6.1f} us {2.0*M*N*K/ms/1e9:6.0f} TF"
    print(line, flush=True)
    del a, w, out
ops.set_option("gemm_store", None)
print("sum (ms):", {v: round(t, 3) for v, t in tot.items()})
