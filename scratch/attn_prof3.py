"""Timeline of the persistent attn_bwd_fused3_kernel from the MAEST_ATTN_PROF build: stamps of workgroup 5, steps 0..127."""
import sys, ctypes, torch, numpy as np
sys.path.insert(0, ".")
from maest_amd import ops, _lib
raw = ctypes.CDLL(sys.argv[1] if len(sys.argv) > 1 else "maest_amd/libmaest_hip_prof.so")
_lib._lib = _lib._bind(raw)
B, N = 256, 290
dev = "cuda"
qkv = torch.randn(B * N, 2304, device=dev).to(torch.bfloat16)
out, lse = ops.attn_fwd(qkv, B, N, 0.125, save_lse=True)
do = torch.randn_like(out)
buf = torch.zeros(12 * 128 * 8, dtype=torch.int64, device=dev)
for _ in range(3):
    ops.attn_bwd(qkv, out, do, lse, B, N, 0.125)
torch.cuda.synchronize()
raw.maest_debug_attn_prof.argtypes = [ctypes.c_void_p]
assert raw.maest_debug_attn_prof(buf.data_ptr()) == 0
ops.attn_bwd(qkv, out, do, lse, B, N, 0.125)
torch.cuda.synchronize()
raw.maest_debug_attn_prof(None)
a = buf.cpu().numpy().reshape(12, 128, 8).astype(np.float64)
nqt = (N + 31) // 32
steps = 120
st = a[:, :steps, :7]
print("total cycles first stamp -> last stamp (wave 0):", int(st[0, steps - 1, 6] - st[0, 0, 0]))
names_key = ["dK/dV store + DMA issue + frag load", "S, dP issued", "softmax + dS write", "dV, dK issued", "vmcnt/lgkmcnt wait", "barrier wait"]
names_aux = ["vmcnt(0) wait", "stats + dq store", "dQ product", "kt_load", "lgkmcnt(0)", "barrier wait"]
for role, waves, names in (("key", list(range(0, 10)), names_key), ("aux", [10, 11], names_aux)):
    d = np.diff(st[waves], axis=-1)                     # [wave, step, 6]
    period = st[waves][:, 1:, 0] - st[waves][:, :-1, 0]
    tmod = np.arange(steps) % nqt
    print(f"--- {role} waves, mean cycles per phase; all steps | first step of an item (t = 0) | last (t = {nqt - 1}) | others")
    for k in range(6):
        sel = lambda m: d[:, m, k].mean()
        print(f"  {names[k]:38s} {d[:, 10:, k].mean():7.0f} | {sel((tmod == 0) & (np.arange(steps) >= 10)):7.0f} | {sel(tmod == nqt - 1):7.0f} | {sel((tmod > 0) & (tmod < nqt - 1) & (np.arange(steps) >= 10)):7.0f}")
    back = st[waves][:, 1:, 0] - st[waves][:, :-1, 6]
    print(f"  (flush + loop back)                    {back.mean():7.0f}")
    print(f"  step period                            {period[:, 10:].mean():7.0f}")
    for w in range(len(waves)):
        print("    wave", waves[w], " ".join(f"{d[w, 10:, k].mean():6.0f}" for k in range(6)))
