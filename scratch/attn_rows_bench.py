# attention forward / backward restricted to the head tokens (last block) against the complete kernels, B = 256
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
for B, N in ((256, 290), (256, 560)):
    qkv = torch.randn(B * N, 2304, device=dev).to(dt)
    out, lse = ops.attn_fwd(qkv, B, N, 0.125, save_lse=True)
    print(f"N={N} fwd complete {bench(lambda: ops.attn_fwd(qkv, B, N, 0.125, save_lse=True)):7.1f} us   head rows {bench(lambda: ops.attn_fwd(qkv, B, N, 0.125, save_lse=True, q_rows=2)):7.1f} us")
    if ops.attn_bwd_rows_supported(dt, N):
        dc = torch.randn(B * 2, 768, device=dev).to(dt)
        dfull = ops.scatter_head_rows(dc, B, N, 2, N); dpart = ops.scatter_head_rows(dc, B, N, 2, 32)
        print(f"N={N} bwd complete {bench(lambda: ops.attn_bwd(qkv, out, dfull, lse, B, N, 0.125)):7.1f} us   head rows {bench(lambda: ops.attn_bwd(qkv, out, dpart, lse, B, N, 0.125, q_rows=2)):7.1f} us")
