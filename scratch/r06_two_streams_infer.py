"""Two half batches on two streams with the NT GEMMs launched NARROW (persistent, W workgroups, a whole CU each), so that the other
stream's LayerNorm / attention kernels find free CUs beside a GEMM: inference bf16, 10 s clips, 256 clips per pass either way."""
import sys, time, torch
sys.path.insert(0, ".")
from maest_amd import get_maest, ops

dev = torch.device("cuda:0")
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
net = get_maest("discogs-maest-10s-pw-129e", pretrained=False, input_t=625, n_classes=400, precision=prec).to(dev).eval()
g = torch.Generator(device=dev).manual_seed(7)
x = torch.randn((256, 1, 96, 626), generator=g, device=dev)
xa, xb = x[:128].contiguous(), x[128:].contiguous()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
IT = 10

def one():
    for _ in range(IT):
        net(x)

def two():
    cur = torch.cuda.current_stream()
    sa.wait_stream(cur); sb.wait_stream(cur)
    for _ in range(IT):
        with torch.cuda.stream(sa):
            net(xa)
        with torch.cuda.stream(sb):
            net(xb)
    cur.wait_stream(sa); cur.wait_stream(sb)

def timed(fn):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter(); fn(); torch.cuda.synchronize()
    return (time.perf_counter() - t) / IT * 1e3

with torch.no_grad():
    ref = net(x)[0].float()
    with torch.cuda.stream(sa):
        ra = net(xa)[0].float()
    torch.cuda.synchronize()
    print("half batch on a side stream vs full batch, max |d logits|: %.3g" % (ra - ref[:128]).abs().max().item())
    for W in (0,):
        ops.set_option("gemm_wgs", W)
        t1 = timed(one); t2 = timed(two); t1b = timed(one); t2b = timed(two)
        print("NT GEMM workgroups %3d: one stream x 256 clips %.2f / %.2f ms; two streams x 128 clips %.2f / %.2f ms" % (W, t1, t1b, t2, t2b), flush=True)
