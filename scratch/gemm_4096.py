"""4096^3 (and neighbours) on random bf16 operands: gemm_nt256o_kernel (forced onto M = 4096 with gemm_min_m = 512), the 8-wave kernel (gemm_variant = 3) and
torch.matmul (hipBLASLt), paired, min of 5 rounds of 20 launches"""
import sys, torch
sys.path.insert(0, ".")
from maest_amd import ops
dev = "cuda"; dt = torch.bfloat16
torch.manual_seed(0)
def bench(fn, n=20):
    for _ in range(3): fn()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for M, N, K in ((4096, 4096, 4096), (8192, 4096, 4096), (8192, 8192, 8192), (16384, 4096, 4096)):
    a = (torch.rand(M, K, device=dev) * 2 - 1).to(dt); w = (torch.rand(N, K, device=dev) * 2 - 1).to(dt)
    out = torch.empty(M, N, device=dev, dtype=dt)
    t = {"one-wave": [], "one-wave persistent": [], "8-wave": [], "hipBLASLt": []}
    for rnd in range(5):
        with ops.options(gemm_min_m=512, gemm_variant=0): t["one-wave"].append(bench(lambda: ops.gemm_nt(a, w, None, out=out)))
        with ops.options(gemm_min_m=512, gemm_variant=0, gemm_wgs=256): t["one-wave persistent"].append(bench(lambda: ops.gemm_nt(a, w, None, out=out)))
        with ops.options(gemm_min_m=512, gemm_variant=3): t["8-wave"].append(bench(lambda: ops.gemm_nt(a, w, None, out=out)))
        t["hipBLASLt"].append(bench(lambda: torch.matmul(a, w.t(), out=out)))
    fl = 2.0 * M * N * K / 1e9
    print(f"M={M} N={N} K={K} (uniform [-1, 1) operands): " + " | ".join(f"{k} {min(v)*1e3:7.1f} us {fl/min(v):6.0f} TF/s" for k, v in t.items()), flush=True)
