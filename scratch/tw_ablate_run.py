"""time gemm_tn256o_kernel in every library under scratch/pw_abl (scratch/tw_ablate.sh): interleaved rounds, best per variant"""
import sys, glob, ctypes, torch
sys.path.insert(0, ".")
from maest_amd import ops, _lib
dev = "cuda"; dt = torch.bfloat16
def bench(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
K = 74240
shapes = [("qkv", 2304, 768), ("proj", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072)]
data = {s: (torch.randn(K, s[1], device=dev).to(dt), torch.randn(K, s[2], device=dev).to(dt), torch.zeros(s[1], s[2], device=dev), torch.zeros(s[1], device=dev)) for s in shapes}
libs = [(p.split("libmaest_")[1][:-3], _lib._bind(ctypes.CDLL(p))) for p in sorted(glob.glob("scratch/pw_abl/libmaest_*.so"))]
res = {(n, s): [] for n, _ in libs for s in shapes}
for rnd in range(3):
    for name, lib in libs:
        _lib._lib = lib
        ops._option_cache.clear()
        with ops.options(gemm_variant=(3 if name.startswith("old") else 0)):
            for s in shapes:
                a, b, o, c = data[s]
                res[(name, s)].append(bench(lambda: ops.gemm_tn(a, b, o, colsum=c, split_k=0)) * 1e3)
for name, _ in libs:
    print(f"{name:>10s}:" + "".join(f"  {s[0]} {min(res[(name, s)]):7.1f} us {2.0 * K * s[1] * s[2] / min(res[(name, s)]) / 1e6:6.0f} TF" for s in shapes), flush=True)
