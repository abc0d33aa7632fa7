import sys, torch, time
sys.path.insert(0, ".")
from maest_amd import ops
dev = "cuda"
M = 256 * 290
def bench(name, fn, flops, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"{name:28s} {ms:8.3f} ms  {flops/ms/1e9:8.1f} TF/s")
dt = torch.bfloat16
def mk(r, c, dtype=dt): return torch.randn(r, c, device=dev).to(dtype)
for (nm, N, K) in [("qkv", 2304, 768), ("proj", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072)]:
    a = mk(M, K); w = mk(N, K); bias = torch.randn(N, device=dev)
    out_bf = torch.empty(M, N, device=dev, dtype=dt)
    bench(f"{nm} fwd none->bf16", lambda: ops.gemm_nt(a, w, bias, out=out_bf), 2.0*M*N*K)
    out32 = torch.empty(M, N, device=dev)
    res = torch.randn(M, N, device=dev)
    bench(f"{nm} fwd residual->f32", lambda: ops.gemm_nt(a, w, bias, out=out32, epi=ops.EPI_RESIDUAL, aux_in=res), 2.0*M*N*K)
    bench(f"{nm} fwd gelu->bf16", lambda: ops.gemm_nt(a, w, bias, out=out_bf, epi=ops.EPI_GELU), 2.0*M*N*K)
    aux = torch.empty(M, N, device=dev, dtype=dt)
    bench(f"{nm} fwd gelu+aux->bf16", lambda: ops.gemm_nt(a, w, bias, out=out_bf, epi=ops.EPI_GELU, aux_out=aux), 2.0*M*N*K)
    bench(f"{nm} mul->bf16", lambda: ops.gemm_nt(a, w, None, out=out_bf, epi=ops.EPI_MUL, aux_in=aux), 2.0*M*N*K)
# wgrad: dW[N,K] = dY^T[N,Mpad] X^T[K,Mpad]
for (nm, N, K) in [("qkv", 2304, 768), ("proj", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072)]:
    at = mk(N, M); bt = mk(K, M)
    dw = torch.zeros(N, K, device=dev)
    import math
    tiles = math.ceil(N/128)*math.ceil(K/128)
    for sk in (1, max(1, 256//tiles), max(1, 1024 // tiles), max(1, 4096//tiles)):
        bench(f"{nm} wgrad splitk={sk}", lambda: ops.gemm_nt(at, bt, None, out=dw, epi=ops.EPI_ATOMIC, split_k=sk), 2.0*M*N*K)
x = mk(M, 3072)
bench("transpose [M,3072] bf16", lambda: ops.transpose(x, M), 0)
x = mk(M, 768)
bench("transpose [M,768] bf16", lambda: ops.transpose(x, M), 0)
g = torch.zeros(3072, device=dev)
x = mk(M, 3072)
bench("colsum [M,3072]", lambda: ops.colsum(x, g), 0)
print("---- TN wgrad (token-major operands)")
for (nm, N, K) in [("qkv", 2304, 768), ("proj", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072)]:
    dy = mk(M, N); xx = mk(M, K)
    dw = torch.zeros(N, K, device=dev); db = torch.zeros(N, device=dev)
    tiles = math.ceil(N/128)*math.ceil(K/128)
    for sk in (0, max(1, 128//tiles*4), max(1, 512 // tiles)):
        bench(f"{nm} TN wgrad+bias splitk={sk}", lambda: ops.gemm_tn(dy, xx, dw, colsum=db, split_k=sk), 2.0*M*N*K)
