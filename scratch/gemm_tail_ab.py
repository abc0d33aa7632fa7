# A/B on one box, interleaved: the last partial round in 128-row tiles (MAEST_OPT_GEMM_TAIL) off / on, for every NT GEMM of a
# training block at the bench shapes; then the K = 768 plain-epilogue GEMM over N (what is special about N = 2304?)
import sys, torch
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
for M in (256 * 290, 128 * 875, 256 * 560):
    print("M =", M)
    tot = [0.0, 0.0]
    for (nm, N, K, epi) in [("qkv", 2304, 768, "none"), ("proj", 768, 768, "none"), ("fc1", 3072, 768, "pair"), ("fc2", 768, 3072, "none"),
                            ("dfc2", 3072, 768, "mul"), ("dfc1", 768, 3072, "none"), ("dproj", 768, 768, "none"), ("dqkv", 768, 2304, "none")]:
        a = mk(M, K); w = mk(N, K); bias = torch.randn(N, device=dev)
        out = torch.empty(M, N, device=dev, dtype=dt); aux = torch.randn(M, N, device=dev).to(dt)
        fn = {"none": lambda: ops.gemm_nt(a, w, bias, out=out),
              "pair": lambda: ops.gemm_nt(a, w, bias, out=out, epi=ops.EPI_GELU, aux_out=aux),
              "mul": lambda: ops.gemm_nt(a, w, None, out=out, epi=ops.EPI_MUL, aux_in=aux)}[epi]
        r = [[], []]
        for rep in range(3):
            for v in (0, 1):
                ops.set_option("gemm_tail", v)
                r[v].append(bench(fn))
        m = [sorted(x)[1] for x in r]
        tot[0] += m[0]; tot[1] += m[1]
        fl = 2.0 * M * N * K
        print(f"  {nm:6s} {epi:5s} tail off {m[0]:7.3f} ms ({fl/m[0]/1e9:7.1f} TF/s) | on {m[1]:7.3f} ms ({fl/m[1]/1e9:7.1f} TF/s)  {100*(m[1]/m[0]-1):+5.1f} %")
        del a, w, out, aux
    print(f"  block total: off {tot[0]:.3f} ms, on {tot[1]:.3f} ms  {100*(tot[1]/tot[0]-1):+.1f} %")
ops.set_option("gemm_tail", 0)
M = 256 * 290
print("K = 768, plain epilogue, over N (tail off); tiles / 256 = rounds")
for N in (768, 1024, 1536, 2048, 2304, 2560, 3072, 4096):
    a = mk(M, 768); w = mk(N, 768); bias = torch.randn(N, device=dev); out = torch.empty(M, N, device=dev, dtype=dt)
    ms = bench(lambda: ops.gemm_nt(a, w, bias, out=out))
    tiles = 290 * N // 256
    print(f"  N {N:5d}: {ms:7.3f} ms {2.0*M*N*768/ms/1e9:7.1f} TF/s   {tiles} tiles = {tiles/256:.2f} rounds; per round {ms/-(-tiles//256)*1e3:6.1f} us")
    del a, w, out
print("M = 65536 (exact rounds at N = 768: 3.0)")
M = 65536
for N in (768, 2304, 3072):
    a = mk(M, 768); w = mk(N, 768); bias = torch.randn(N, device=dev); out = torch.empty(M, N, device=dev, dtype=dt)
    ms = bench(lambda: ops.gemm_nt(a, w, bias, out=out))
    tiles = 256 * N // 256
    print(f"  N {N:5d}: {ms:7.3f} ms {2.0*M*N*768/ms/1e9:7.1f} TF/s   {tiles} tiles = {tiles/256:.2f} rounds; per round {ms/-(-tiles//256)*1e3:6.1f} us")
