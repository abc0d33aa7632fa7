"""Where do the small fill / copy kernels of a training step come from?  One step under torch.profiler with stacks; prints the python
frames of every aten::zeros / fill_ / zero_ / copy_ call (count per call site)."""
import sys, collections, argparse, torch
sys.path.insert(0, ".")
import bench
args = argparse.Namespace(precision="bf16", complete_last_block=False, serial_kernels=False, no_fold_delta=False, hip_graph=False, force_collective=False)
dev = torch.device("cuda", 0)
case = bench.build_case(args, dev, 0, 1, "train", 626, 256, 30)
for _ in range(3): case["step"]()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    case["step"]()
    torch.cuda.synchronize()
sites = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::zeros", "aten::fill_", "aten::zero_", "aten::copy_", "aten::zeros_like", "aten::full"):
        st = [f for f in (ev.stack or []) if "maest_amd" in f or "bench.py" in f or "optim" in f]
        sites[(ev.name, st[0] if st else "?")] += 1
for (name, where), n in sites.most_common(40):
    print(f"{n:4d}  {name:16s} {where}")
