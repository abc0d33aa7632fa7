import sys, time, torch
sys.path.insert(0, ".")
from maest_amd import get_maest
dev = "cuda"
net = get_maest("discogs-maest-10s-pw-129e", pretrained=False, precision="bf16").to(dev).eval()
for B in (1, 2, 4, 8, 16):
    x = torch.randn(B, 96, 626, device=dev)
    res = {}
    for mode in ("eager", "graph"):
        net.enable_hip_graph(mode == "graph")
        with torch.no_grad():
            for _ in range(5): net(x)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            n = 50
            for _ in range(n): net(x)
            torch.cuda.synchronize(); res[mode] = (time.perf_counter() - t0) / n * 1e3
    print(f"B={B:3d}: eager {res['eager']:.3f} ms  graph {res['graph']:.3f} ms  ({B/res['graph']*1e3:.0f} clips/s)")
