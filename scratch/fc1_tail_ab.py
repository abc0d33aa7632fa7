"""fc1 with the GELU + GELU' pair epilogue (M = 74240, N = 3072, K = 768): the one-wave-per-SIMD 256 x 256 kernel (one workgroup per CU: the epilogue's
VALU work runs with the matrix pipe idle) against the 128-row-tile kernel (two workgroups per CU: one's epilogue can run under the other's main loop)."""
import sys, torch
sys.path.insert(0, ".")
from maest_amd import ops
dev = "cuda"; dt = torch.bfloat16
torch.manual_seed(0)
def mk(r, c, dtype=dt, s=1.0): return (torch.randn(r, c, device=dev) * s).to(dtype)
def bench(fn, n=10):
    for _ in range(2): fn()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
M, N, K = 74240, 3072, 768
a = mk(M, K); w = mk(N, K, s=0.05); bias = torch.randn(N, device=dev)
out = torch.empty(M, N, device=dev, dtype=dt); aux = torch.empty(M, N, device=dev, dtype=dt)
forms = {"256-row tiles, one wg per tile": dict(), "256-row persistent (256 wgs)": dict(gemm_wgs=256), "128-row tiles everywhere (gemm_tail=2)": dict(gemm_tail=2),
         "8-wave 256 kernel (variant 3)": dict(gemm_variant=3)}
for rnd in range(2):
    for nm, o in forms.items():
        with ops.options(**o):
            t = bench(lambda: ops.gemm_nt(a, w, bias, out=out, epi=ops.EPI_GELU, aux_out=aux))
            t0 = bench(lambda: ops.gemm_nt(a, w, bias, out=out))
        print(f"{nm:45s} gelu pair {t*1e3:7.1f} us ({2.0*M*N*K/t/1e9:6.1f} TF/s)   plain {t0*1e3:7.1f} us", flush=True)
