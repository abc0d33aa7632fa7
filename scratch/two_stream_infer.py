"""configs[1] inference as TWO half batches on two streams, the persistent NT GEMMs of each half limited to W workgroups (MAEST_GEMM_WGS):
does an HBM-bound kernel of one half (LayerNorm, attention stream) fill the CUs the other half's GEMM leaves?  Prints ms per 256-clip step."""
import os, sys, time, torch
sys.path.insert(0, ".")
from maest_amd import get_maest, ops
dev = torch.device("cuda", 0)
net = get_maest("discogs-maest-10s-pw-129e", pretrained=False, precision="bf16").to(dev).eval()
x = torch.randn((256, 1, 96, 626), device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def one():
    with torch.no_grad():
        return net(x)[0]
def two(wgs):
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.no_grad(), ops.options(gemm_wgs=wgs):
        with torch.cuda.stream(s1):
            a = net(x[:128])[0]
        with torch.cuda.stream(s2):
            b = net(x[128:])[0]
    cur.wait_stream(s1); cur.wait_stream(s2)
    return a, b
def interleaved(wgs):
    """the same with the two halves' kernels enqueued alternately block by block is not expressible through net(): threads instead"""
    import threading
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    def run(s, xs, out, i):
        with torch.no_grad(), torch.cuda.stream(s):
            out[i] = net(xs)[0]
    out = [None, None]
    with ops.options(gemm_wgs=wgs):
        t1 = threading.Thread(target=run, args=(s1, x[:128], out, 0)); t2 = threading.Thread(target=run, args=(s2, x[128:], out, 1))
        t1.start(); t2.start(); t1.join(); t2.join()
    cur.wait_stream(s1); cur.wait_stream(s2)
    return out
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
print("one stream, batch 256:", round(bench(one), 3), "ms")
for w in (256, 192, 160, 128, 96):
    print(f"two streams, halves of 128, gemm_wgs {w}: sequential enqueue {bench(lambda: two(w)):.3f} ms, threaded enqueue {bench(lambda: interleaved(w)):.3f} ms")
print("one stream, batch 256:", round(bench(one), 3), "ms")
