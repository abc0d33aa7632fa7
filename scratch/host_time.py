# host-side time of each part of the training step (no synchronisation inside the step) against the synchronised step time,
# with and without the forced one-rank collective: is the step host-bound anywhere?
import os, sys, time, socket
import torch
sys.path.insert(0, ".")
force = "--force" in sys.argv
from maest_amd import get_maest
from maest_amd.dist import GradReducer, init_from_env
from maest_amd.module import Module
if force:
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    init_from_env(force=True)
dev = torch.device("cuda", 0)
net = get_maest("passt_s_swa_p16_128_ap476", pretrained=False, input_t=625, s_patchout_t=30, precision="bf16").to(dev).train()
mod = Module(net=net, mixup_alpha=0.3)
opt = mod.get_optimizer()
red = None
if force:
    red = GradReducer(net.named_parameters(), skip=("head_dist.weight", "head_dist.bias"), force_collective=True)
    net._grad_sink = red
x = torch.randn((256, 1, 96, 626), device=dev); y = (torch.rand((256, 400), device=dev) < 0.006).float()
def step(rec=None):
    t0 = time.perf_counter()
    if red: red.reset()
    loss = mod.training_step((x, None, y), 0)
    t1 = time.perf_counter()
    loss.backward()
    t2 = time.perf_counter()
    if red: red.finish()
    t3 = time.perf_counter()
    opt.step()
    if not red: opt.zero_grad(set_to_none=True)
    t4 = time.perf_counter()
    if rec is not None: rec.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3))
for _ in range(3): step()
torch.cuda.synchronize()
rec = []
T0 = time.perf_counter()
for _ in range(10): step(rec)
T1 = time.perf_counter()
torch.cuda.synchronize()
T2 = time.perf_counter()
import numpy as np
r = np.array(rec) * 1e3
print(("forced collective" if force else "plain"), "host ms per step: fwd %.2f  bwd %.2f  finish %.2f  opt %.2f  | host loop %.2f ms/step, synced %.2f ms/step"
      % (*r.mean(0), (T1 - T0) * 100, (T2 - T0) * 100))
