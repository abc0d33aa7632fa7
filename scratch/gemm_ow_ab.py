"""One-wave-per-SIMD NT GEMM (gemm_nt_ow.hip, default for bf16) against the 8-wave kernel (gemm_variant = 3): equality of the
results (same products, different summation order inside a k-step? no: the same order -- compared with a tolerance and
bit-for-bit) and paired timing, min of interleaved rounds."""
import sys, torch
sys.path.insert(0, ".")
from maest_amd import ops
dev = "cuda"; dt = torch.bfloat16
torch.manual_seed(0)
def mk(r, c, dtype=dt, s=1.0): return (torch.randn(r, c, device=dev) * s).to(dtype)
def run(variant, fn):
    with ops.options(gemm_variant=variant):
        return fn()
# ---- correctness at a ragged M, every epilogue
M, N, K = 256 * 3 + 77, 768, 768
a = mk(M, K); w = mk(N, K, s=0.05); bias = torch.randn(N, device=dev)
res = torch.randn(M, N, device=dev); mul = mk(M, N)
ref = a.float() @ w.float().t() + bias
for name, kw in [("none->bf16", dict(out_dtype=dt)), ("none->f32", dict(out_dtype=torch.float32)),
                 ("gelu->bf16", dict(out_dtype=dt, epi=ops.EPI_GELU)),
                 ("resid->f32", dict(out_dtype=torch.float32, epi=ops.EPI_RESIDUAL, aux_in=res)),
                 ("mul->bf16", dict(out_dtype=dt, epi=ops.EPI_MUL, aux_in=mul))]:
    o_new = run(0, lambda: ops.gemm_nt(a, w, bias, **kw))
    o_old = run(3, lambda: ops.gemm_nt(a, w, bias, **kw))
    print(f"{name:11s} new vs old: max abs diff {(o_new.float() - o_old.float()).abs().max().item():.3e}  bit-equal {torch.equal(o_new, o_old)}")
o = run(0, lambda: ops.gemm_nt(a, w, bias, out_dtype=torch.float32))
print("none->f32 vs fp32 matmul: max abs err", (o - ref).abs().max().item(), " ref max", ref.abs().max().item())
aux_n = torch.empty(M, N, device=dev, dtype=dt); aux_o = torch.empty(M, N, device=dev, dtype=dt)
o_new = run(0, lambda: ops.gemm_nt(a, w, bias, out_dtype=dt, epi=ops.EPI_GELU, aux_out=aux_n))
o_old = run(3, lambda: ops.gemm_nt(a, w, bias, out_dtype=dt, epi=ops.EPI_GELU, aux_out=aux_o))
print("gelu+aux   bit-equal", torch.equal(o_new, o_old), torch.equal(aux_n, aux_o))
# ---- timing
def bench(fn, n=10):
    for _ in range(2): fn()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
shapes = [("big", 4096, 4096, 4096), ("big2", 65536, 4096, 4096)]
for M in (74240, 65536):
    shapes += [(f"qkv{M}", M, 2304, 768), (f"proj{M}", M, 768, 768), (f"fc1{M}", M, 3072, 768), (f"fc2{M}", M, 768, 3072)]
for nm, M, N, K in shapes:
    a = mk(M, K); w = mk(N, K, s=0.05); bias = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev, dtype=dt)
    t = {0: [], 3: []}
    for rnd in range(3):
        for v in (0, 3):
            t[v].append(run(v, lambda: bench(lambda: ops.gemm_nt(a, w, bias, out=out))))
    lib = bench(lambda: torch.matmul(a, w.t()))
    fl = 2.0 * M * N * K / 1e9
    print(f"{nm:10s} M={M:6d} N={N:5d} K={K:5d}: one-wave {min(t[0]):7.3f} ms {fl/min(t[0]):7.1f} TF/s | 8-wave {min(t[3]):7.3f} ms {fl/min(t[3]):7.1f} TF/s | "
          f"hipBLASLt {lib:7.3f} ms {fl/lib:7.1f} TF/s", flush=True)
    del a, w, out
