"""bf16 attention backward, the forms side by side (interleaved rounds): usage attn_bwd_forms.py [rounds]
   1 = two-kernel (dK/dV + two-block dQ), 5 = two-kernel with the one-block dQ kernel, 0 = default (fused where it applies)"""
import sys, statistics, torch
sys.path.insert(0, ".")
from maest_amd import ops
dev = "cuda"; dt = torch.bfloat16
def bench(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
for B, N in ((128, 875), (256, 290), (256, 560), (64, 1685)):
    x = torch.randn(B * N, 2304, device=dev)
    x[:, :768] *= 0.125 * 1.4426950408889634
    qkv = x.to(dt)
    out, lse = ops.attn_fwd(qkv, B, N, 0.125, save_lse=True, q_prescaled=True)
    do = torch.randn_like(out)
    delta = (out.float() * do.float()).reshape(B, N, 12, 64).sum(-1).permute(0, 2, 1).contiguous()
    res = {}
    for _ in range(rounds):
        for form in (0, 1, 5):
            with ops.options(attn_bwd=form):
                res.setdefault(form, []).append(bench(lambda: ops.attn_bwd(qkv, None, do, lse, B, N, 0.125, delta=delta, q_prescaled=True)))
    print(f"B={B} N={N}: " + "   ".join(f"form {f}: min {min(v):7.1f} med {statistics.median(v):7.1f} us" for f, v in res.items()), flush=True)
