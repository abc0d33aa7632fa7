"""Are the compiler-scheduled attention kernels paced by the matrix instruction's energy?  The product library against one whose mma_chunk<bf16> issues two
16x16x32 MFMAs in place of each 32x32x16 (-DMAEST_CRUDE16: wrong results, same flops and operands); alternating, min of 3."""
import sys, ctypes, torch
sys.path.insert(0, ".")
from maest_amd import ops, _lib
dev = "cuda"; dt = torch.bfloat16
def bench(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
new = _lib.load(); crude = _lib._bind(ctypes.CDLL("scratch/pw_abl/libmaest_attn16.so"))
for B, N in ((256, 290), (128, 875), (256, 560)):
    qkv = (torch.randn(B * N, 2304, device=dev) * 0.5).to(dt)
    out, lse = ops.attn_fwd(qkv, B, N, 0.125, save_lse=True)
    do = torch.randn_like(out)
    t = {}
    for rnd in range(3):
        for name, lib in (("32x32", new), ("crude16", crude)):
            _lib._lib = lib; ops._option_cache.clear()
            with ops.options(attn_fwd=2):
                t.setdefault((name, "fwd4w"), []).append(bench(lambda: ops.attn_fwd(qkv, B, N, 0.125)))
            t.setdefault((name, "bwd"), []).append(bench(lambda: ops.attn_bwd(qkv, out, do, lse, B, N, 0.125)))
    _lib._lib = new; ops._option_cache.clear()
    print(f"B={B} N={N}: four-wave fwd {min(t[('32x32','fwd4w')]):7.1f} -> {min(t[('crude16','fwd4w')]):7.1f} us | bwd {min(t[('32x32','bwd')]):7.1f} -> {min(t[('crude16','bwd')]):7.1f} us", flush=True)
