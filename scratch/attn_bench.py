import sys, torch
sys.path.insert(0, ".")
from maest_amd import ops, _lib
import ctypes
if len(sys.argv) > 1:
    _lib._lib = _lib._bind(ctypes.CDLL(sys.argv[1]))   # A/B: another build of the library
    print("lib:", sys.argv[1])
dev = "cuda"; dt = torch.bfloat16
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
import inspect
for (B, N) in [(256, 290), (256, 560)]:
    qkv = torch.randn(B * N, 2304, device=dev).to(dt)
    r = ops.attn_fwd(qkv, B, N, 0.125, save_lse=True) if "save_lse" in inspect.signature(ops.attn_fwd).parameters else ops.attn_fwd(qkv, B, N, 0.125)
    out, lse = r if isinstance(r, tuple) else (r, None)
    do = torch.randn_like(out)
    f = bench(lambda: ops.attn_fwd(qkv, B, N, 0.125))
    line = f"B={B} N={N}: fwd {f*1e3:7.1f} us ({4.0*N*N*64*12*B/f/1e9:6.1f} TF/s)"
    if lse is not None:
        b = bench(lambda: ops.attn_bwd(qkv, out, do, lse, B, N, 0.125))
        line += f"  bwd {b*1e3:7.1f} us ({10.0*N*N*64*12*B/b/1e9:6.1f} TF/s)"
    print(line)
