# odd-shape fuzz of the eval forward (fp32 parity mode) against the oracle
import sys, numpy as np, torch
sys.path.insert(0, ".")
from maest_amd import get_maest
from oracle import maest_oracle as O
dev = "cuda"
sd = O.make_state_dict(625, seed=3)
net = get_maest("discogs-maest-10s-pw-129e", pretrained=False, precision="fp32"); net.load_state_dict(sd); net = net.to(dev).eval()
rng = np.random.Generator(np.random.PCG64(0))
worst = 0
for B, T in [(1, 16), (1, 17), (1, 25), (2, 26), (3, 36), (1, 101), (5, 333), (2, 625), (1, 626), (7, 59)]:
    x = torch.from_numpy(rng.standard_normal((B, 96, T), dtype=np.float32))
    want, wf = O.forward(x, sd, (96, 625), melspectrogram_input=True) if False else O.forward(x.unsqueeze(1), sd, (96, 625))
    with torch.no_grad():
        got, gf = net(x.unsqueeze(1).to(dev))
    e = ((got.cpu() - want).abs().max() / want.abs().max()).item()
    for blk in (0, 11):
        _, emb = net(x.unsqueeze(1).to(dev), transformer_block=blk)
        _, we = O.forward(x.unsqueeze(1), sd, (96, 625), transformer_block=blk)
        e = max(e, ((emb.cpu() - we).abs().max() / we.abs().max()).item())
    worst = max(worst, e)
    print(f"B={B} T={T}: tokens {2 + 9 * ((T - 16) // 10 + 1)}  rel err {e:.2e}")
print("worst", worst)
assert worst < 1e-3
