"""A/B of the two-kernel attention backward: DMA-fed tiles (default at N > 320) against register-staged tiles (attn_bwd = 4)."""
import sys, torch
sys.path.insert(0, ".")
from maest_amd import ops
torch.manual_seed(0)
for B, N in ((128, 875), (256, 560), (32, 1685), (256, 290)):
    qkv = (torch.randn(B * N, 2304, device="cuda") * 0.5).to(torch.bfloat16)
    dout = (torch.randn(B * N, 768, device="cuda") * 0.1).to(torch.bfloat16)
    out, lse = ops.attn_fwd(qkv, B, N, 0.125, save_lse=True)
    res = {}
    for rnd in range(3):
        for mode in (1, 4):
            with ops.options(attn_bwd=mode):
                for _ in range(2): g = ops.attn_bwd(qkv, out, dout, lse, B, N, 0.125)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10): g = ops.attn_bwd(qkv, out, dout, lse, B, N, 0.125)
                e1.record(); torch.cuda.synchronize()
                res.setdefault(mode, []).append(e0.elapsed_time(e1) / 10 * 1e3)
                res.setdefault(("g", mode), g)
    eq = torch.equal(res[("g", 1)], res[("g", 4)])
    print(f"B={B} N={N}: dma {min(res[1]):.1f} us  reg {min(res[4]):.1f} us  bit-equal {eq}", flush=True)
