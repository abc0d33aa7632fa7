"""One-wave-per-SIMD TN GEMM (gemm_tn_ow.hip, default for bf16) against the 8-wave kernel (gemm_variant = 3) and torch: results and
paired timing (min of interleaved rounds).  out[M][N] += a[K][M]^T b[K][N]."""
import sys, torch
sys.path.insert(0, ".")
from maest_amd import ops
dev = "cuda"; dt = torch.bfloat16
torch.manual_seed(0)
def run(variant, fn):
    with ops.options(gemm_variant=variant):
        return fn()
def tn(a, b, cs=True):
    out = torch.zeros(a.shape[1], b.shape[1], device=dev); c = torch.zeros(a.shape[1], device=dev) if cs else None
    ops.gemm_tn(a, b, out, colsum=c, split_k=0)
    return out, c
for K, M, N in ((2048, 768, 768), (74240, 768, 768), (9280, 2304, 768), (160, 256, 256), (32 * 37, 512, 256)):
    a = torch.randn(K, M, device=dev).to(dt); b = torch.randn(K, N, device=dev).to(dt)
    ref = a.float().t() @ b.float(); refc = a.float().sum(0)
    o_n, c_n = run(4, lambda: tn(a, b)) if M * N < 8 * 65536 else run(0, lambda: tn(a, b))
    o_o, c_o = run(3, lambda: tn(a, b))
    sc = ref.abs().max().item()
    print(f"K={K} M={M} N={N}: new vs ref {(o_n - ref).abs().max().item() / sc:.2e}  old vs ref {(o_o - ref).abs().max().item() / sc:.2e}  "
          f"colsum new {(c_n - refc).abs().max().item() / refc.abs().max().item():.2e} old {(c_o - refc).abs().max().item() / refc.abs().max().item():.2e}", flush=True)
def bench(fn, n=10):
    for _ in range(2): fn()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
K = 74240
for nm, M, N in (("qkv", 2304, 768), ("proj", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072)):
    a = torch.randn(K, M, device=dev).to(dt); b = torch.randn(K, N, device=dev).to(dt)
    out = torch.zeros(M, N, device=dev); cs = torch.zeros(M, device=dev)
    t = {0: [], 3: []}
    for rnd in range(3):
        for v in (0, 3):
            t[v].append(run(v, lambda: bench(lambda: ops.gemm_tn(a, b, out, colsum=cs, split_k=0))))
    fl = 2.0 * M * N * K / 1e9
    print(f"{nm:5s} out={M:5d} in={N:5d}: one-wave {min(t[0]):7.3f} ms {fl/min(t[0]):7.1f} TF/s | 8-wave {min(t[3]):7.3f} ms {fl/min(t[3]):7.1f} TF/s", flush=True)
