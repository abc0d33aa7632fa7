"""Timeline of attn_bwd_fused2_kernel from the MAEST_ATTN_PROF build (scratch/attn_prof.sh): shader-clock stamps of every
wave of the workgroups with blockIdx % 256 == 5, per query tile; prints the mean phase durations per wave role."""
import sys, ctypes, torch, numpy as np
sys.path.insert(0, ".")
from maest_amd import ops, _lib
path = sys.argv[1] if len(sys.argv) > 1 else "maest_amd/libmaest_hip_prof.so"
raw = ctypes.CDLL(path)
_lib._lib = _lib._bind(raw)
B, N = 256, 290
dev = "cuda"
qkv = torch.randn(B * N, 2304, device=dev).to(torch.bfloat16)
out, lse = ops.attn_fwd(qkv, B, N, 0.125, save_lse=True)
do = torch.randn_like(out)
nslot = B * 12 // 256
buf = torch.zeros(nslot * 12 * 16 * 8, dtype=torch.int64, device=dev)
for _ in range(3):
    ops.attn_bwd(qkv, out, do, lse, B, N, 0.125)
torch.cuda.synchronize()
raw.maest_debug_attn_prof.argtypes = [ctypes.c_void_p]
assert raw.maest_debug_attn_prof(buf.data_ptr()) == 0
ops.attn_bwd(qkv, out, do, lse, B, N, 0.125)
torch.cuda.synchronize()
raw.maest_debug_attn_prof(None)
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ops.attn_bwd(qkv, out, do, lse, B, N, 0.125)
e1.record(); torch.cuda.synchronize()
print(f"launch time (instrumented build, stamps off): {e0.elapsed_time(e1) * 100:.1f} us")
a = buf.cpu().numpy().reshape(nslot, 12, 16, 8).astype(np.float64)
nkw, nqt = (N + 31) // 32, (N + 31) // 32
life = a[:, :, 15, 1] - a[:, :, 15, 0]
print("workgroup life (cycles), per slot (wave 0):", life[:, 0].astype(int))
start = a[:, 0, 15, 0]
print("workgroup start offsets (cycles) rel. slot 0:", (start - start[0]).astype(int))
names_key = ["top->S,dP issued", "softmax+dS write", "dV,dK issued", "lgkmcnt(0)", "barrier wait", "(loop back)"]
names_aux = ["vmcnt(0) wait", "stat/dma/stat/dq store", "dQ product", "lgkmcnt(0)", "barrier wait", "(loop back)"]
for role, waves, names in (("key", range(0, nkw), names_key), ("aux", range(nkw, nkw + 2), names_aux)):
    st = a[1:, :, :nqt, :6][:, list(waves)]            # slots 1.. (steady state), [slot, wave, tile, stamp]
    d = np.diff(st, axis=-1)                            # 5 phases
    nxt = st[:, :, 1:, 0] - st[:, :, :-1, 5]            # loop back
    print(f"--- {role} waves: mean cycles per phase over slots 1.., all waves, tiles 1..{nqt - 2}")
    for k in range(5):
        print(f"  {names[k]:28s} {d[:, :, 1:-1, k].mean():8.0f}   (min {d[:, :, 1:-1, k].min():6.0f}  max {d[:, :, 1:-1, k].max():6.0f})")
    print(f"  {names[5]:28s} {nxt.mean():8.0f}")
    per_tile = st[:, :, 1:, 0] - st[:, :, :-1, 0]
    print(f"  tile period                  {per_tile.mean():8.0f}")
    print("  per wave mean [S/dP, softmax, dV/dK, lgkm, barrier]:")
    for w in range(d.shape[1]):
        print("    wave", list(waves)[w], " ".join(f"{d[:, w, 1:-1, k].mean():6.0f}" for k in range(5)))
# prologue / epilogue of the workgroup
t_first = a[1:, 0, 0, 0] - a[1:, 0, 15, 0]
t_last = a[1:, 0, 15, 1] - a[1:, 0, nqt - 1, 5]
print(f"prologue (start -> first tile top, wave 0): {t_first.mean():.0f} cycles; epilogue (last barrier -> end): {t_last.mean():.0f}")
