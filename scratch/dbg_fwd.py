import sys, torch, numpy as np
sys.path.insert(0, ".")
from maest_amd import get_maest, ops
from oracle import maest_oracle as O
m = get_maest("discogs-maest-10s-pw-129e", pretrained=False, precision="fp32")
m.load_state_dict(O.make_state_dict(625)); m = m.cuda().eval()
x3 = torch.randn(2, 96, 626, device="cuda")
toff, tok = m._resolve_tokens(9, 62)
print("tok", tok.shape, tok.dtype, tok[:3].tolist(), tok[-1].tolist(), toff)
tok = tok.cuda()
cols = ops.patch_im2col(x3, tok, torch.float32); torch.cuda.synchronize(); print("im2col ok", cols.shape)
W = m._engine.w
patches = ops.gemm_nt(cols, W.get(m.patch_embed.proj.weight, torch.float32), m.patch_embed.proj.bias, out_dtype=torch.float32); torch.cuda.synchronize(); print("gemm ok")
Tt = m.time_new_pos_embed.shape[-1]
x = ops.token_assemble(patches, m.cls_token.reshape(-1), m.dist_token.reshape(-1), m.new_pos_embed.reshape(2, 768), m.freq_new_pos_embed.reshape(768, -1), m.time_new_pos_embed.reshape(768, Tt), toff, tok, 2); torch.cuda.synchronize(); print("assemble ok")
