"""time gemm_nt256d_kernel in every library under scratch/pw_abl (scratch/owd_ablate.sh) + gemm_nt256o_kernel of the first one (gemm_defer = 0):
interleaved rounds, best per variant; persistent form (256 workgroups)"""
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
shapes = [(74240, 2304, 768), (74240, 768, 768), (74240, 768, 3072), (143360, 2304, 768), (143360, 768, 768)]
data = {}
for (M, N, K) in shapes:
    data[(M, N, K)] = ((torch.randn(M, K, device=dev)).to(dt), (torch.randn(N, K, device=dev) * 0.05).to(dt), torch.randn(N, device=dev),
                       torch.empty(M, N, device=dev, dtype=dt))
libs = [(p.split("libmaest_")[1][:-3], _lib._bind(ctypes.CDLL(p))) for p in sorted(glob.glob("scratch/pw_abl/libmaest_*.so"))]
res = {(n, s): [] for n, _ in libs for s in shapes}
for rnd in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    for name, lib in libs:
        _lib._lib = lib
        ops._option_cache.clear()
        with ops.options(gemm_tail=0, gemm_wgs=256):
            for s in shapes:
                a, w, b, o = data[s]
                res[(name, s)].append(bench(lambda: ops.gemm_nt(a, w, b, out=o)) * 1e3)
for name, _ in libs:
    line = f"{name:>14s}:"
    for s in shapes:
        v = res[(name, s)]
        line += f"  {s[0]}x{s[1]}x{s[2]} {min(v):7.1f} us {2.0 * s[0] * s[1] * s[2] / min(v) / 1e6:6.0f} TF"
    print(line, flush=True)
