# full-size check (B = 256, N = 290 / 281): attention restricted to the head's tokens against the complete kernels
import sys, torch
sys.path.insert(0, ".")
from maest_amd import ops
dev = "cuda"; dt = torch.bfloat16
for B, N in ((256, 290), (256, 281), (37, 320)):
    torch.manual_seed(B + N)
    qkv = torch.randn(B * N, 2304, device=dev).to(dt)
    full, lse = ops.attn_fwd(qkv, B, N, 0.125, save_lse=True)
    part, lse_p = ops.attn_fwd(qkv, B, N, 0.125, save_lse=True, q_rows=2)
    nv = 32
    assert torch.equal(part.view(B, N, 768)[:, :nv], full.view(B, N, 768)[:, :nv]) and torch.equal(lse_p[:, :, :nv], lse[:, :, :nv])
    dc = torch.randn(B * 2, 768, device=dev).to(dt)
    want = ops.attn_bwd(qkv, full, ops.scatter_head_rows(dc, B, N, 2, N), lse, B, N, 0.125)
    got = ops.attn_bwd(qkv, part, ops.scatter_head_rows(dc, B, N, 2, nv), lse_p, B, N, 0.125, q_rows=2)
    d = (got.float() - want.float()).abs().max().item(); s = want.float().abs().max().item()
    zq = got.view(B, N, 2304)[:, 2:, :768].abs().max().item()
    print(f"B={B} N={N}: max |restricted - complete| = {d:.3e} (scale {s:.3e}), max |dQ| beyond the head rows = {zq}")
    assert d <= 2e-2 * s and zq == 0.0
print("ok")
