import sys, torch
sys.path.insert(0, ".")
from maest_amd import ops
dev = "cuda"
B, Fg, Tt = 1, 9, 62
tok = torch.tensor([[0, 0], [0, 1], [1, 0]], dtype=torch.int32, device=dev)
P = 3
patches = torch.ones(B * P, 768, device=dev)
z = lambda *s: torch.zeros(*s, device=dev)
print("calling", flush=True)
x0 = ops.token_assemble(patches, z(768), z(768), z(2, 768), z(768, Fg), z(768, Tt), 0, tok, B)
torch.cuda.synchronize()
print("ok", x0.shape, x0.sum().item(), flush=True)
