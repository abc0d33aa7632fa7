"""attn_fwd_pw_kernel timeline: needs scratch/pw_abl/libmaest_prof.so (scratch/pw_ablate.sh "prof:-DPW_PROF").  Prints, for both waves of
workgroup 5, the shader-clock distance between consecutive stamps of the first items (attn_fwd_pw.hip: PW_STAMP)."""
import sys, ctypes, torch
sys.path.insert(0, ".")
from maest_amd import ops, _lib
lib = ctypes.CDLL("scratch/pw_abl/libmaest_%s.so" % (sys.argv[3] if len(sys.argv) > 3 else "prof"))
_lib._lib = _lib._bind(lib)
B, N = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (256, 560)
qkv = torch.randn(B * N, 2304, device="cuda").to(torch.bfloat16)
buf = torch.zeros(2 * 512 + 64, dtype=torch.int64, device="cuda")
with ops.options(attn_fwd=3):
    for _ in range(3): ops.attn_fwd(qkv, B, N, 0.125)
    torch.cuda.synchronize()
    lib.maest_debug_pw_prof.argtypes = [ctypes.c_void_p]
    assert lib.maest_debug_pw_prof(buf.data_ptr()) == 0
    ops.attn_fwd(qkv, B, N, 0.125)
    torch.cuda.synchronize()
    lib.maest_debug_pw_prof(None)
print("XCC_ID register of workgroups 0..63:", [int(x) & 0xf for x in buf.cpu()[1024:]]); t = buf.cpu()[:1024].reshape(2, 512)
T = (N + 63) // 64
per_item = 2 + 4 * T + 4          # stamps per item: start, prologue, 4 per tile, drain/wait/take/store
for w in range(2):
    v = [int(x) for x in t[w] if x != 0]
    d = [v[i + 1] - v[i] for i in range(len(v) - 1)]
    print(f"wave {w}: {len(v)} stamps, {per_item} per item; first item spans {v[per_item] - v[0] if len(v) > per_item else -1} cycles")
    for it in range(min(3, len(v) // per_item)):
        seg = d[it * per_item: (it + 1) * per_item]
        print(f"  item {it}: prologue {seg[0]}")
        for tl in range(T):
            r = seg[1 + 4 * tl: 5 + 4 * tl]
            print(f"    tile {tl}: region0 {r[0]:5d}  barrier {r[1]:5d}  region1 {r[2]:5d}  region2 {r[3]:5d}   sum {sum(r)}")
        tail = seg[1 + 4 * T:]
        print(f"    tail (drain, Q wait, Q take, stores->next item start): {tail}")
