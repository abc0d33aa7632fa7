import sys, torch
sys.path.insert(0, ".")
from maest_amd import ops
from tests import kernel_cases as KC
B, N = 1, 20
qkv = KC.rnd((B*N, 2304), 1)
q_, k_, v_ = qkv.reshape(B, N, 3, 12, 64).permute(2, 0, 3, 1, 4)
att = ((q_ @ k_.transpose(-2, -1)) * 0.125).softmax(-1)
out, lse = ops.attn_fwd(qkv.cuda(), B, N, 0.125, save_lse=True)
ref, ref_lse = KC._attn_ref(qkv, B, N, 0.125)
print("lse err", (lse.cpu()-ref_lse).abs().max().item(), "out err", (out.cpu()-ref).abs().max().item())
res = []
for k0 in range(N):
    q2 = qkv.clone()
    v = q2[:, 1536:].reshape(N, 12, 64); v[:] = 0; v[k0] = 1.0
    out = ops.attn_fwd(q2.cuda(), B, N, 0.125).cpu()
    d = (att[0,0,0,:] - out[0,0]).abs()
    res.append((int(d.argmin()), round(float(d.min()),6), round(float(out[0,0]),5)))
print(res)
print("att row0", att[0,0,0,:])
q2 = qkv.clone(); v = q2[:, 1536:].reshape(N, 12, 64); v[:] = torch.arange(64, dtype=torch.float32)[None,None,:]
out = ops.attn_fwd(q2.cuda(), B, N, 0.125).cpu()
print("V=d:", out[0,:16], out[7, 64:72])
