import sys, torch, ctypes
sys.path.insert(0, ".")
from maest_amd import ops, _lib
if len(sys.argv) > 1:
    _lib._lib = _lib._bind(ctypes.CDLL(sys.argv[1])); print("lib:", sys.argv[1])
dev = "cuda"; dt = torch.bfloat16
def bench(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for B, N in ((128, 875), (64, 1685)):
    qkv = torch.randn(B * N, 2304, device=dev).to(dt)
    out, lse = ops.attn_fwd(qkv, B, N, 0.125, save_lse=True)
    do = torch.randn_like(out)
    print(f"B={B} N={N}: fwd {bench(lambda: ops.attn_fwd(qkv, B, N, 0.125)):8.1f} us  bwd {bench(lambda: ops.attn_bwd(qkv, out, do, lse, B, N, 0.125)):8.1f} us")
