# FETCH_SIZE / WRITE_SIZE calibration on known byte counts (run under rocprofv3 --pmc FETCH_SIZE, then WRITE_SIZE):
#   copy      : torch copy of 512 MiB fp32  (reads 512 MiB, writes 512 MiB)
#   ln_fwd    : layernorm_fwd of [74240, 768] fp32 -> bf16 (reads 228.1 MB, writes 114.0 MB)
#   gemm_n256 : gemm_nt [74240, 768] x [256, 768]^T bf16 -> bf16: every A line is needed by exactly ONE workgroup
#               (reads 114.0 MB + 0.4 MB, writes 38.0 MB): LDS-DMA full-line access pattern of gemm_nt256w_kernel
#   gemm_qkv  : the qkv shape N = 2304 (algorithmic reads 114.0 + 3.5 MB, writes 342.1 MB)
import sys, torch
sys.path.insert(0, ".")
from maest_amd import ops
dev = "cuda"
M = 74240
a = torch.randn(M, 768, device=dev).bfloat16()
w256 = torch.randn(256, 768, device=dev).bfloat16()
wqkv = torch.randn(2304, 768, device=dev).bfloat16()
o256 = torch.empty(M, 256, device=dev, dtype=torch.bfloat16)
oqkv = torch.empty(M, 2304, device=dev, dtype=torch.bfloat16)
x = torch.randn(M, 768, device=dev)
g = torch.ones(768, device=dev); b = torch.zeros(768, device=dev)
src = torch.randn(128 * 1024 * 1024, device=dev); dst = torch.empty_like(src)
junk = torch.empty(256 * 1024 * 1024, device=dev)
for rep in range(3):
    junk.zero_()                       # flush L2 / MALL between the measured launches
    dst.copy_(src)
    junk.zero_()
    ops.layernorm_fwd(x, g, b, 1e-6, torch.bfloat16)
    junk.zero_()
    ops.gemm_nt(a, w256, None, out=o256)
    junk.zero_()
    ops.gemm_nt(a, wqkv, None, out=oqkv)
torch.cuda.synchronize()
print("done")
