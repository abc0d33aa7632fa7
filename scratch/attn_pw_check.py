"""attn_fwd_pw_kernel (MAEST_ATTN_FWD=3) on the GPU: against an fp32 torch reference on the same rounded operands and against the
four-wave LDS-DMA kernel (MAEST_ATTN_FWD=2), then timing of both at the production shapes."""
import sys, torch
sys.path.insert(0, ".")
from maest_amd import ops
dev = "cuda"

def ref_attn(qkv, B, N, scale):
    x = qkv.float().reshape(B, N, 3, 12, 64).permute(2, 0, 3, 1, 4)
    q, k, v = x[0], x[1], x[2]
    s = (q @ k.transpose(-1, -2)) * scale
    lse = torch.logsumexp(s, -1)
    o = torch.softmax(s, -1) @ v
    return o.permute(0, 2, 1, 3).reshape(B * N, 768), lse

def check(B, N, spike=False, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    qkv = torch.randn(B * N, 2304, generator=g)
    if spike:
        for key, qrow, f in ((min(N - 1, 70), 3, 6.0), (N - 1, 5, 4.0), (N // 2, N - 2, 8.0)):
            qkv[key, 768:768 + 64] = qkv[qrow, 0:64] * f
    qkv = qkv.to(torch.bfloat16).to(dev)
    ref, rl = ref_attn(qkv, B, N, 0.125)
    res = {}
    for mode in (3, 2):
        with ops.options(attn_fwd=mode):
            o, l = ops.attn_fwd(qkv, B, N, 0.125, save_lse=True)
        torch.cuda.synchronize()
        res[mode] = ((o.float() - ref).abs().max().item(), (l - rl).abs().max().item(), bool(torch.isfinite(o.float()).all()))
    # run to run: the persistent kernel has no atomics, results must repeat bit for bit
    with ops.options(attn_fwd=3):
        o1, l1 = ops.attn_fwd(qkv, B, N, 0.125, save_lse=True)
        o2, l2 = ops.attn_fwd(qkv, B, N, 0.125, save_lse=True)
    rep = torch.equal(o1, o2) and torch.equal(l1, l2)
    ok = res[3][0] < 2e-2 and res[3][1] < 2e-2 and res[3][2] and rep
    print(f"B={B:3d} N={N:5d} spike={int(spike)}: pw |dO| {res[3][0]:.4f} |dlse| {res[3][1]:.4f}  dma |dO| {res[2][0]:.4f} |dlse| {res[2][1]:.4f}  repeat={rep} {'OK' if ok else 'FAIL'}", flush=True)
    return ok

def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

allok = True
for (B, N, sp) in [(1, 40, False), (2, 75, False), (1, 130, True), (3, 200, True), (2, 321, False), (2, 560, False), (2, 560, True), (1, 875, True),
                   (1, 1685, False), (13, 875, False), (32, 560, False), (5, 1685, True)]:
    allok &= check(B, N, sp)
print("ALL OK" if allok else "SOME FAILED", flush=True)
if "--bench" in sys.argv:
    for (B, N) in [(256, 560), (128, 875), (64, 1685), (256, 290)]:
        qkv = torch.randn(B * N, 2304, device=dev).to(torch.bfloat16)
        line = f"B={B} N={N}:"
        for rnd in range(2):
            for mode in (2, 3):
                with ops.options(attn_fwd=mode):
                    t = bench(lambda: ops.attn_fwd(qkv, B, N, 0.125))
                line += f"  mode{mode} {t*1e3:7.1f} us ({4.0*N*N*64*12*B/t/1e9:6.1f} TF/s)"
        print(line, flush=True)
