import sys, torch, collections
sys.path.insert(0, ".")
import numpy as np
from maest_amd import get_maest, ops, _lib
from maest_amd.module import Module
dev = "cuda"
net = get_maest("passt_s_swa_p16_128_ap476", pretrained=False, input_t=625, s_patchout_t=30, precision="bf16").to(dev).train()
mod = Module(net=net)
opt = mod.get_optimizer()
import os
if os.environ.get("MAEST_SERIAL"): net._engine.overlap_wgrad = False     # each event pair then times one kernel alone
B = 256
x = torch.randn(B, 1, 96, 626, device=dev); y = (torch.rand(B, 400, device=dev) < 0.006).float()
recs = []
orig = ops._timed_call
def hook(name, work, *args, _entry=None):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); _lib.call(_entry or name, *args); e1.record()
    tag = name
    if name in ("maest_gemm_nt", "maest_gemm_nt_small"):
        tag = f"nt M={args[8]} N={args[9]} K={args[10]} epi={args[12]} out={args[7]} aux={'y' if args[14] is not None else 'n'}"
    elif name == "maest_gemm_tn":
        tag = f"tn M={args[7]} N={args[8]} K={args[9]} sk={args[11]}"
    recs.append((tag, e0, e1, work))
def step():
    loss = mod.training_step((x, None, y), 0); loss.backward(); opt.step(); opt.zero_grad()
for _ in range(2): step()
ops._timed_call = hook
torch.cuda.synchronize()
import time; t0 = time.perf_counter()
for _ in range(3): step()
torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 3 * 1e3
agg = collections.OrderedDict()
for tag, e0, e1, w in recs:
    d = agg.setdefault(tag, [0, 0.0, 0.0]); d[0] += 1; d[1] += e0.elapsed_time(e1); d[2] += w
tot = 0
for tag, (n, ms, w) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    tot += ms / 3
    print(f"{tag:64s} n/step {n/3:5.1f}  ms/step {ms/3:7.3f}  avg {ms/n:7.3f} ms  {w/ms/1e9 if w else 0:7.1f} TF/s")
print("sum of timed kernels ms/step", tot, "wall ms/step (instrumented)", wall)
