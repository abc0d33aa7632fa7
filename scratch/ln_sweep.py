import sys, torch
sys.path.insert(0, ".")
from maest_amd import ops
dev = "cuda"
M = 74240
def bench(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
x = torch.randn(M, 768, device=dev); dy = torch.randn(M, 768, device=dev).bfloat16(); dres = torch.randn(M, 768, device=dev)
g = torch.ones(768, device=dev); b = torch.zeros(768, device=dev)
y, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-6, torch.bfloat16, save_stats=True)
dg = torch.zeros(768, device=dev); db = torch.zeros(768, device=dev)
junk = torch.empty(512 * 1024 * 1024 // 4, device=dev)
def z(): junk.zero_()
tz = bench(z)
for blocks in (256, 512, 1024, 2048, 4096, 18560):
    ops.set_option("ln_bwd_blocks", blocks)
    def f():
        junk.zero_()
        ops.layernorm_bwd(dy, x, g, mean, rstd, dres, dg, db, lp_dtype=torch.bfloat16)
    t = bench(f) - tz
    print(f"layernorm_bwd blocks={blocks:6d}: {t*1e3:7.1f} us  ({(M*768*(2+4+4+4+2))/t/1e9:.2f} TB/s)")
ops.set_option("ln_bwd_blocks", None)
d = torch.randn(M, 768, device=dev).bfloat16()
def f2():
    junk.zero_(); ops.add_layernorm_fwd(x, d, g, b, 1e-6, torch.bfloat16, save_stats=True)
def f1():
    junk.zero_(); ops.layernorm_fwd(x, g, b, 1e-6, torch.bfloat16, save_stats=True)
t1, t2 = bench(f1) - tz, bench(f2) - tz
print(f"layernorm_fwd {t1*1e3:.1f} us ({M*768*6/t1/1e9:.2f} TB/s)   add_layernorm_fwd {t2*1e3:.1f} us ({M*768*12/t2/1e9:.2f} TB/s)")
