"""Training step (configs[2]: 256 clips, 10 s, patchout 30) as two 128-clip halves on two streams vs one 256-clip pass."""
import sys, time, torch
sys.path.insert(0, ".")
from maest_amd import get_maest, ops
from maest_amd.module import Module

dev = torch.device("cuda:0")
net = get_maest("passt_s_swa_p16_128_ap476", pretrained=False, input_t=625, n_classes=400, s_patchout_t=30, distilled_type="mean", precision="bf16").to(dev)
net.train()
mod = Module(net=net, mixup_alpha=0.3)
opt = mod.get_optimizer(net.parameters())
g = torch.Generator(device=dev).manual_seed(7)
x = torch.randn((256, 1, 96, 626), generator=g, device=dev)
y = (torch.rand((256, 400), generator=g, device=dev) < 2.5 / 400).float()
xa, xb, ya, yb = x[:128].contiguous(), x[128:].contiguous(), y[:128].contiguous(), y[128:].contiguous()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
IT = 10

def one():
    for _ in range(IT):
        loss = mod.training_step((x, None, y), 0)
        loss.backward(); opt.step(); opt.zero_grad(set_to_none=True)

def two():
    cur = torch.cuda.current_stream()
    for _ in range(IT):
        sa.wait_stream(cur); sb.wait_stream(cur)
        with torch.cuda.stream(sa):
            la = mod.training_step((xa, None, ya), 0)
        with torch.cuda.stream(sb):
            lb = mod.training_step((xb, None, yb), 0)
        cur.wait_stream(sa); cur.wait_stream(sb)
        la.record_stream(cur); lb.record_stream(cur)
        loss = (la + lb) * 0.5
        loss.backward(); opt.step(); opt.zero_grad(set_to_none=True)

def timed(fn):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter(); fn(); torch.cuda.synchronize()
    return (time.perf_counter() - t) / IT * 1e3

for W in (0, 224):
    ops.set_option("gemm_wgs", W)
    t1 = timed(one); t2 = timed(two); t1b = timed(one); t2b = timed(two)
    print("NT GEMM workgroups %3d: one stream x 256 clips %.2f / %.2f ms; two streams x 128 clips %.2f / %.2f ms" % (W, t1, t1b, t2, t2b), flush=True)
