"""Looks for one-off stalls in the training step: N steps, a HIP event and a host timestamp behind each; prints the steps whose GPU-side
or host-side duration exceeds 1.3x the median, plus allocator statistics (cudaMalloc retries / segment counts) before and after."""
import sys, time, statistics, argparse, gc, torch
sys.path.insert(0, ".")
import bench
args = argparse.Namespace(precision="bf16", complete_last_block=False, serial_kernels=False, no_fold_delta=False, hip_graph=False, force_collective=False)
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
freeze = len(sys.argv) > 2 and sys.argv[2] == "freeze"      # gc.collect() + gc.freeze() behind the warm-up, as bench.py does
case = bench.build_case(args, dev, 0, 1, "train", 626, 256, 30)
step = case["step"]
for _ in range(5): step()
torch.cuda.synchronize()
if freeze:
    gc.collect()
    gc.freeze()
st0 = torch.cuda.memory_stats()
evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
host = []
gcs = []
def cb(phase, info):
    if phase == "start": gcs.append([time.perf_counter(), info["generation"], None])
    else: gcs[-1][2] = time.perf_counter()
gc.callbacks.append(cb)
evs[0].record()
t0 = time.perf_counter()
for i in range(n):
    step()
    evs[i + 1].record()
    host.append(time.perf_counter())
torch.cuda.synchronize()
gc.callbacks.remove(cb)
gpu = [evs[i].elapsed_time(evs[i + 1]) for i in range(n)]
hst = [(host[i] - (host[i - 1] if i else t0)) * 1e3 for i in range(n)]
mg, mh = statistics.median(gpu), statistics.median(hst)
print(f"{n} steps: GPU median {mg:.2f} ms, host-enqueue median {mh:.2f} ms, total wall {(host[-1] - t0) * 1e3 / n:.2f} ms/step")
for i in range(n):
    if gpu[i] > 1.3 * mg or hst[i] > 1.3 * mh + 5:
        print(f"  step {i}: GPU {gpu[i]:.1f} ms, host {hst[i]:.1f} ms")
st1 = torch.cuda.memory_stats()
for k in ("num_alloc_retries", "num_ooms", "segment.all.allocated", "reserved_bytes.all.peak", "num_device_alloc", "num_device_free"):
    print(f"  {k}: {st0.get(k)} -> {st1.get(k)}")
big = [(b - a) * 1e3 for a, g, b in gcs if b and (b - a) > 0.005]
print(f"  gc collections during the loop: {len(gcs)} (gen2: {sum(1 for g in gcs if g[1] == 2)}); longer than 5 ms: {[round(x, 1) for x in big]}")
