"""persistent fused attention backward (MAEST_OPT_ATTN_BWD = 0) against the one-workgroup-per-item form (= 3): same math in the
same order -> bit-equal dqkv expected; several items per workgroup (B * 12 > 256) so that the item boundary is crossed."""
import sys, torch
sys.path.insert(0, ".")
from maest_amd import ops
dev = "cuda"
torch.manual_seed(0)
for (B, N) in [(2, 290), (30, 290), (64, 281), (43, 257), (256, 290), (256, 320)]:
    qkv = torch.randn(B * N, 2304, device=dev).to(torch.bfloat16)
    out, lse = ops.attn_fwd(qkv, B, N, 0.125, save_lse=True)
    do = torch.randn_like(out)
    ops.set_option("attn_bwd", 3)
    ref = ops.attn_bwd(qkv, out, do, lse, B, N, 0.125)
    ops.set_option("attn_bwd", 0)
    worst = 0.0
    for rep in range(3):
        got = ops.attn_bwd(qkv, out, do, lse, B, N, 0.125)
        torch.cuda.synchronize()
        d = (got.float() - ref.float()).abs().max().item()
        worst = max(worst, d)
    nz = (got != ref).sum().item()
    print(f"B={B} N={N}: max |fused3 - fused2| = {worst:.3e} (ref max {ref.float().abs().max().item():.3e}), differing elements {nz}", flush=True)
