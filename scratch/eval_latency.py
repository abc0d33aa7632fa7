"""Default-precision (bf16x3) evaluation latency at small batches: model.eval()(mel[B, 96, 626]), ms per call (median of 20 after 5 warm-ups)."""
import sys, time, statistics, torch
sys.path.insert(0, ".")
from maest_amd import get_maest
dev = torch.device("cuda", 0)
net = get_maest("discogs-maest-10s-pw-129e", pretrained=False).to(dev).eval()
for B in (1, 2, 4, 8, 16, 32):
    x = torch.randn(B, 96, 626, device=dev)
    ts = []
    with torch.no_grad():
        for i in range(25):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            net(x.clone())
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"B={B:3d}: {statistics.median(ts[5:]):7.2f} ms per call  ({B / statistics.median(ts[5:]) * 1e3:7.1f} clips/s)", flush=True)
# ... and replayed from a captured HIP graph (model.enable_hip_graph())
net.enable_hip_graph()
for B in (1, 8):
    x = torch.randn(B, 96, 626, device=dev)
    ts = []
    with torch.no_grad():
        for i in range(25):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            net(x.clone())
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"B={B:3d} (graph replay): {statistics.median(ts[5:]):7.2f} ms per call", flush=True)
