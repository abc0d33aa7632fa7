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
B, N = 256, 290
qkv = torch.randn(B * N, 2304, device=dev).to(dt)
out, lse = ops.attn_fwd(qkv, B, N, 0.125, save_lse=True)
do = torch.randn_like(out)
names = {0: "fused full", 1: "two-kernel", 2: "no dq jobs", 3: "no staging", 4: "no key-wave compute", 5: "no K prologue",
         6: "only prologue+barriers (no dq, staging, key compute)", 7: "barriers only"}
for rep in range(2):
    for v, nm in names.items():
        ops.set_option("attn_bwd", v)
        t = bench(lambda: ops.attn_bwd(qkv, out, do, lse, B, N, 0.125))
        if rep: print(f"attn_bwd={v} {nm:55s} {t*1e3:8.1f} us")
ops.set_option("attn_bwd", 0)
