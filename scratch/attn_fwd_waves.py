"""attn_fwd_dma_kernel<NW>: waves (32-query blocks) per workgroup, NW in {4, 5, 6, 8}: bit-equality with NW = 4 and timing per N.
Interleaved rounds on one box (profiles/r03_attn_fwd_waves.txt)."""
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


NWS = (4, 5, 6, 8)
shapes = [(256, 290), (256, 281), (256, 560), (128, 875), (64, 1685), (256, 129), (256, 64), (8, 290), (1, 560)]
for (B, N) in shapes:
    torch.manual_seed(N)
    qkv = torch.randn(B * N, 2304, device=dev).to(dt)
    ref = None
    eq = []
    for nw in NWS:
        with ops.options(attn_fwd_waves=nw):
            out, lse = ops.attn_fwd(qkv, B, N, 0.125, save_lse=True)
        if ref is None:
            ref = (out.clone(), lse.clone())
        eq.append(bool(torch.equal(out, ref[0]) and torch.equal(lse, ref[1])))
    t = {nw: [] for nw in NWS}
    for rnd in range(3):
        for nw in NWS:
            with ops.options(attn_fwd_waves=nw):
                t[nw].append(bench(lambda: ops.attn_fwd(qkv, B, N, 0.125, save_lse=True)))
    med = {nw: sorted(t[nw])[1] for nw in NWS}
    print(f"B={B:4d} N={N:5d}  " + "  ".join(f"NW={nw}: {med[nw]:7.1f} us" for nw in NWS) + f"   bit-equal to NW=4: {eq}", flush=True)
