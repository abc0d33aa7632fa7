"""model.eval()(x) with eval_streams = 1 vs 2 (maest_amd/maest.py: MAEST._eval_forward), same box, alternating; outputs compared bit for bit."""
import sys, time, torch
sys.path.insert(0, ".")
from maest_amd import get_maest

dev = torch.device("cuda:0")
IT = 10

def timed(fn):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter(); fn(); torch.cuda.synchronize()
    return (time.perf_counter() - t) / IT * 1e3

def case(arch, img_t, T, B, prec, force=False):
    net = get_maest(arch, pretrained=False, input_t=img_t, n_classes=400, precision=prec).to(dev).eval()
    if force:
        net.EVAL_SPLIT_ROWS = 1
    g = torch.Generator(device=dev).manual_seed(7)
    x = torch.randn((B, 1, 96, T), generator=g, device=dev)
    def loop():
        for _ in range(IT):
            net(x)
    with torch.no_grad():
        res = {}
        for rep in range(2):
            for n in (1, 2):
                net.eval_streams = n
                out = net(x)[0].clone()
                res.setdefault(n, []).append(timed(loop))
                if n == 1: ref = out
                else: same = torch.equal(out, ref)
    rows = B * (2 + net._tok_cache[next(iter(net._tok_cache))][1].shape[0])
    print("%-28s %-7s B=%3d rows=%6d%s: one stream %.2f / %.2f ms, two streams %.2f / %.2f ms (%+.1f %%), outputs bit-equal: %s" % (
        arch, prec, B, rows, " (forced)" if force else "", res[1][0], res[1][1], res[2][0], res[2][1],
        100.0 * (min(res[2]) / min(res[1]) - 1.0), same), flush=True)
    del net, x
    torch.cuda.empty_cache()

for prec in ("bf16", "bf16x3", "fp16"):
    case("discogs-maest-10s-pw-129e", 625, 626, 256, prec)
case("discogs-maest-30s-pw-73e-ts", 1875, 1876, 64, "bf16")
case("discogs-maest-30s-pw-73e-ts", 1875, 1876, 64, "bf16x3")
case("discogs-maest-10s-pw-129e", 625, 626, 192, "bf16")
case("discogs-maest-10s-pw-129e", 625, 626, 128, "bf16", force=True)
case("discogs-maest-10s-pw-129e", 625, 626, 64, "bf16", force=True)
case("discogs-maest-30s-pw-73e-ts", 1875, 1876, 32, "bf16", force=True)
