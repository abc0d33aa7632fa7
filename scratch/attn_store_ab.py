"""16-byte row pieces (v_permlane32_swap, cdna_hip_programming.md T21) instead of 8-byte stores in the attention kernels' store tails:
old library (argv[1]) against the in-tree build on one box, interleaved, bit-equality of the outputs checked."""
import sys, ctypes, torch
sys.path.insert(0, ".")
from maest_amd import ops, _lib
new = _lib.load()
old = _lib._bind(ctypes.CDLL(sys.argv[1]))
dev = "cuda"; dt = torch.bfloat16
def use(l): _lib._lib = l
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (B, N) in [(256, 290), (256, 281), (256, 560), (128, 875), (64, 1685)]:
    torch.manual_seed(N)
    qkv = torch.randn(B * N, 2304, device=dev).to(dt)
    res = {}
    for name, l in (("old", old), ("new", new)):
        use(l)
        out, lse = ops.attn_fwd(qkv, B, N, 0.125, save_lse=True)
        do = torch.randn(B * N, 768, device=dev, generator=torch.Generator(dev).manual_seed(1)).to(dt)
        dq = ops.attn_bwd(qkv, out, do, lse, B, N, 0.125)
        res[name] = (out, lse, dq)
    same = all(torch.equal(a, b) for a, b in zip(res["old"], res["new"]))
    out, lse, _ = res["new"]
    tf = {"old": [], "new": []}; tb = {"old": [], "new": []}
    for rnd in range(3):
        for name, l in (("old", old), ("new", new)):
            use(l)
            tf[name].append(bench(lambda: ops.attn_fwd(qkv, B, N, 0.125, save_lse=True)))
            tb[name].append(bench(lambda: ops.attn_bwd(qkv, out, do, lse, B, N, 0.125)))
    m = lambda v: sorted(v)[1]
    print(f"B={B:4d} N={N:5d}  fwd old {m(tf['old']):7.1f} new {m(tf['new']):7.1f} us ({(m(tf['new'])/m(tf['old'])-1)*100:+5.1f} %)   "
          f"bwd old {m(tb['old']):7.1f} new {m(tb['new']):7.1f} us ({(m(tb['new'])/m(tb['old'])-1)*100:+5.1f} %)   bit-equal: {same}", flush=True)
use(new)
