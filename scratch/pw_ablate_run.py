"""time attn_fwd_pw_kernel in every library under scratch/pw_abl (scratch/pw_ablate.sh): interleaved rounds, best and median per variant"""
import sys, glob, ctypes, statistics, torch
sys.path.insert(0, ".")
from maest_amd import ops, _lib
dev = "cuda"
QS = True     # the model's contract (MAEST_BF16_QS): q columns pre-scaled
def bench(fn, n=15):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
shapes = [(256, 560), (128, 875)]
def make(s):
    x = torch.randn(s[0] * s[1], 2304, device=dev)
    if QS: x[:, :768] *= 0.125 * 1.4426950408889634      # q' = scale * log2(e) * q: the scores of the unscaled case
    return x.to(torch.bfloat16)
data = {s: make(s) for s in shapes}
libs = [(p.split("libmaest_")[1][:-3], _lib._bind(ctypes.CDLL(p))) for p in sorted(glob.glob("scratch/pw_abl/libmaest_*.so"))]
res = {(n, s): [] for n, _ in libs for s in shapes}
for rnd in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    for name, lib in libs:
        _lib._lib = lib
        with ops.options(attn_fwd=3):
            for (B, N) in shapes:
                res[(name, (B, N))].append(bench(lambda: ops.attn_fwd(data[(B, N)], B, N, 0.125, q_prescaled=QS)) * 1e3)
for name, _ in libs:
    line = f"{name:>14s}:"
    for s in shapes:
        v = res[(name, s)]
        line += f"   N={s[1]} min {min(v):6.1f} med {statistics.median(v):6.1f} us"
    print(line, flush=True)
