# NT GEMM first-round phase stagger sweep (MAEST_OPT_GEMM_STAGGER, clocks per K stage and class step), every ViT shape
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
vals = [0, 150, 300, 450, 600, 900]
tot = {v: 0.0 for v in vals}
for (nm, N, K, epi) in [("qkv", 2304, 768, "none"), ("proj", 768, 768, "res"), ("fc1", 3072, 768, "gelu"), ("fc2", 768, 3072, "res"),
                        ("dfc2", 3072, 768, "mul"), ("dfc1", 768, 3072, "none"), ("dproj", 768, 768, "none"), ("dqkv", 768, 2304, "none")]:
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
    line = f"{nm:6s} N={N:4d} K={K:4d} {epi:5s}"
    for rep in range(2):          # two interleaved rounds
        for v in vals:
            ops.set_option("gemm_stagger", v)
            ms = bench(fn)
            if rep == 1:
                tot[v] += ms
                line += f" | s={v}: {ms*1e3:6.1f} us {2.0*M*N*K/ms/1e9:6.0f} TF"
    print(line, flush=True)
    del a, w, out
print("sum over the 8 shapes (ms):", {v: round(t, 3) for v, t in tot.items()})
