# does running two half-batches on two streams hide kernel tails?  (two independent model instances)
import sys, time, torch
sys.path.insert(0, ".")
from maest_amd import get_maest
from maest_amd.module import Module
dev = "cuda"
def make(B):
    net = get_maest("passt_s_swa_p16_128_ap476", pretrained=False, input_t=625, s_patchout_t=30, precision="bf16").to(dev).train()
    mod = Module(net=net); opt = mod.get_optimizer()
    x = torch.randn(B, 1, 96, 626, device=dev); y = (torch.rand(B, 400, device=dev) < 0.006).float()
    def step():
        loss = mod.training_step((x, None, y), 0); loss.backward(); opt.step(); opt.zero_grad()
    return step
def timeit(fn, n=8):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
one = make(256)
t1 = timeit(one)
print(f"1 x B=256, one stream : {t1*1e3:.2f} ms/step  {256/t1:.0f} clips/s")
a, b = make(128), make(128)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def both():
    with torch.cuda.stream(s1): a()
    with torch.cuda.stream(s2): b()
t2 = timeit(both)
print(f"2 x B=128, two streams: {t2*1e3:.2f} ms/pair  {256/t2:.0f} clips/s")
ta = timeit(a)
print(f"1 x B=128, one stream : {ta*1e3:.2f} ms/step  {128/ta:.0f} clips/s")
