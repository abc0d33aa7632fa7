"""attention forward at the training shape (B = 256, N = 290 / 281): the four-wave LDS-DMA kernel (default below N = 321) against the persistent
one-wave-per-SIMD kernel forced onto it (attn_fwd = 3: two work items of 192 rows per (batch, head), 75 % useful); alternating, min of 3."""
import sys, torch
sys.path.insert(0, ".")
from maest_amd import ops
dev = "cuda"
def bench(fn, n=20):
    for _ in range(3): fn()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for B, N in ((256, 290), (256, 281), (256, 320), (128, 290)):
    qkv = (torch.randn(B * N, 2304, device=dev) * 0.5).to(torch.bfloat16)
    t = {}
    outs = {}
    for rnd in range(3):
        for form in (0, 3, 2):
            with ops.options(attn_fwd=form):
                t.setdefault(form, []).append(bench(lambda: ops.attn_fwd(qkv, B, N, 0.125, save_lse=True, q_prescaled=True)))
                outs[form] = ops.attn_fwd(qkv, B, N, 0.125, save_lse=True, q_prescaled=True)
    fl = 4.0 * N * N * 64 * B * 12 / 1e9
    d = (outs[0][0].float() - outs[3][0].float()).abs().max().item()
    print(f"B={B} N={N}: default {min(t[0])*1e3:6.1f} us ({fl/min(t[0]):5.0f} TF) | persistent (3) {min(t[3])*1e3:6.1f} us ({fl/min(t[3]):5.0f} TF) | four-wave forced (2) {min(t[2])*1e3:6.1f} us | max |out diff| {d:.2e}", flush=True)
