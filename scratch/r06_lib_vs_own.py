"""The vendor library (torch.matmul -> hipBLASLt, bf16, fp32 accumulate, plain bf16 output) against gemm_nt256o_kernel / gemm_tn256o_kernel on
the SAME operands at the model's shapes (measurement only: the product never calls a library GEMM).
  python scratch/r06_lib_vs_own.py time            -> paired timings, alternating arms, min of 3 rounds (no profiler)
  python scratch/r06_lib_vs_own.py lib|own  (under rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE: scratch/r06_lib_vs_own.sh)
"""
import json, sys, torch
sys.path.insert(0, ".")
from maest_amd import ops
dev = "cuda"; dt = torch.bfloat16
arm = sys.argv[1]
torch.manual_seed(0)
def mk(r, c, s=1.0): return (torch.randn(r, c, device=dev) * s).to(dt)
def bench(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
NT = [("qkv", 2304, 768), ("proj", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072), ("dqkv", 768, 2304), ("dfc1", 768, 3072), ("dfc2", 3072, 768)]
TN = [("wqkv", 2304, 768), ("wproj", 768, 768), ("wfc1", 3072, 768), ("wfc2", 768, 3072)]
SHAPES = [(M, nm, N, K, "nt") for M in (74240, 112000, 143360) for nm, N, K in NT] + [(4096, "sq4096", 4096, 4096, "nt"), (8192, "sq8192", 8192, 8192, "nt")] + \
         [(74240, nm, No, Ki, "tn") for nm, No, Ki in TN]
NPROF = 5
plan = []
for M, nm, N, K, kind in SHAPES:
    if kind == "nt":
        a = mk(M, K); w = mk(N, K, 0.05); out = torch.empty(M, N, device=dev, dtype=dt); wt = w.t()
        lib = lambda: torch.matmul(a, wt, out=out)
        def own():
            with ops.options(gemm_wgs=256, gemm_tail=0):
                ops.gemm_nt(a, w, None, out=out)
        fl = 2.0 * M * N * K
    else:
        dy = mk(M, N); x = mk(M, K); dw = torch.zeros(N, K, device=dev); db = torch.zeros(N, device=dev)
        outb = torch.empty(N, K, device=dev, dtype=dt); dyt = dy.t()
        lib = lambda: torch.matmul(dyt, x, out=outb)
        own = lambda: ops.gemm_tn(dy, x, dw, colsum=db, split_k=0)
        fl = 2.0 * M * N * K
    if arm == "time":
        tl, to = [], []
        for r in range(3):
            tl.append(bench(lib)); to.append(bench(own))
        print(f"{kind} M={M:6d} {nm:6s} N={N:5d} K={K:5d}: library {min(tl)*1e3:7.1f} us {fl/min(tl)/1e9:7.1f} TF/s | own {min(to)*1e3:7.1f} us {fl/min(to)/1e9:7.1f} TF/s | own/lib {min(tl)/min(to):.3f}x", flush=True)
    else:
        fn = lib if arm == "lib" else own
        for _ in range(NPROF): fn()
        torch.cuda.synchronize()
        plan.append({"M": M, "name": nm, "N": N, "K": K, "kind": kind, "launches": NPROF, "flops": fl})
if arm != "time":
    json.dump(plan, open(f"gpurun_out/r06_lib/plan_{arm}.json", "w"))
