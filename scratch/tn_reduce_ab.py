"""Split-K combine of gemm_tn256_kernel: partial tiles through a workspace + tn256_reduce_kernel (tn_reduce=1, default) against fp32
atomics (tn_reduce=0), the wgrad shapes of a block at the bench's token count, interleaved on one box."""
import sys, torch
sys.path.insert(0, ".")
from maest_amd import ops
dev = "cuda"; dt = torch.bfloat16
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for K in (74240, 112000):
    for name, M, N in (("qkv", 2304, 768), ("proj", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072)):
        a = torch.randn(K, M, device=dev).to(dt); b = torch.randn(K, N, device=dev).to(dt)
        out = torch.zeros(M, N, device=dev); cs = torch.zeros(M, device=dev)
        t = {0: [], 1: []}
        for rnd in range(3):
            for v in (0, 1):
                with ops.options(tn_reduce=v):
                    t[v].append(bench(lambda: ops.gemm_tn(a, b, out, colsum=cs, split_k=0)))
        m = lambda v: sorted(v)[1]
        print(f"tokens {K:6d} {name:5s} [{M:4d} x {N:4d}]  atomics {m(t[0]):7.1f} us   workspace + reduce {m(t[1]):7.1f} us  ({(m(t[1])/m(t[0])-1)*100:+5.1f} %)   "
              f"workspace {256 * 256 * 4 * 256 // ((M // 256) * (N // 256)) * ((M // 256) * (N // 256)) / 2**20:.0f} MiB at most", flush=True)
