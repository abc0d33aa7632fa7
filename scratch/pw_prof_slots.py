"""per-slot stamps (build with -DPW_PROF -DPW_PROF_SLOTS): tile 1 of the first item (PAR = 1): the 16 slots of regions 0, 1, 2"""
import sys, ctypes, torch
sys.path.insert(0, ".")
from maest_amd import ops, _lib
lib = ctypes.CDLL("scratch/pw_abl/libmaest_%s.so" % sys.argv[1])
_lib._lib = _lib._bind(lib)
B, N = 256, 560
qkv = torch.randn(B * N, 2304, device="cuda").to(torch.bfloat16)
buf = torch.zeros(2 * 512 + 64, dtype=torch.int64, device="cuda")
with ops.options(attn_fwd=3):
    for _ in range(3): ops.attn_fwd(qkv, B, N, 0.125)
    torch.cuda.synchronize()
    lib.maest_debug_pw_prof.argtypes = [ctypes.c_void_p]
    lib.maest_debug_pw_prof(buf.data_ptr())
    ops.attn_fwd(qkv, B, N, 0.125); torch.cuda.synchronize()
    lib.maest_debug_pw_prof(None)
v = [int(x) for x in buf.cpu()[:512] if x != 0]
d = [v[i + 1] - v[i] for i in range(len(v) - 1)]
# stamps: item start, prologue, tile 0: 4 stamps, tile 1 (PAR 1): 16 + 1, 1, 16 + 1, 16 + 1 ...
i0 = 2 + 4 - 1          # index of the diff that ends at tile 1's first slot stamp
print("tile 1 region 0 slots:", d[i0:i0 + 16], " tail", d[i0 + 16])
print("barrier:", d[i0 + 17])
print("tile 1 region 1 slots:", d[i0 + 18:i0 + 34], " tail", d[i0 + 34])
print("tile 1 region 2 slots:", d[i0 + 35:i0 + 51], " tail", d[i0 + 51])
