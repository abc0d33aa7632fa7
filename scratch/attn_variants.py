# attention backward variants side by side (MAEST_OPT_ATTN_BWD: 0 fused DMA-fed, 1 two-kernel, 2 fused register-fed)
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
    return e0.elapsed_time(e1) / n
for (B, N) in [(256, 290), (256, 281), (64, 129)]:
    qkv = torch.randn(B * N, 2304, device=dev).to(dt)
    out, lse = ops.attn_fwd(qkv, B, N, 0.125, save_lse=True)
    do = torch.randn_like(out)
    res = {}
    line = f"B={B} N={N}: fwd {bench(lambda: ops.attn_fwd(qkv, B, N, 0.125, save_lse=True))*1e3:7.1f} us |"
    for rep in range(2):
        for v, nm in ((0, "fused dma"), (1, "two-kernel"), (2, "fused regs")):
            ops.set_option("attn_bwd", v)
            t = bench(lambda: ops.attn_bwd(qkv, out, do, lse, B, N, 0.125))
            if rep:
                res[v] = ops.attn_bwd(qkv, out, do, lse, B, N, 0.125).float()
                line += f" bwd {nm}: {t*1e3:7.1f} us"
    ops.set_option("attn_bwd", None)
    err = max((res[0] - res[1]).abs().max().item(), (res[2] - res[1]).abs().max().item()) / res[1].abs().max().item()
    print(line, f"| max rel diff between variants {err:.2e}")
