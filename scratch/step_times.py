"""per-step wall times of two training cases built one after the other in one process (what bench.py's loader_case does after the headline case)"""
import sys, time, argparse, torch
sys.path.insert(0, ".")
import bench
args = argparse.Namespace(precision="bf16", hip_graph=False, complete_last_block=False, serial_kernels=False, no_fold_delta=False, force_collective=False,
                          ranks_share_gpu=False)
dev = torch.device("cuda:0")
for T in (626, 625, 626):
    try:
        case = bench.build_case(args, dev, 0, 1, "train", T, 256, 30)
    except AttributeError as e:
        print("args missing:", e); raise
    ts = []
    for i in range(14):
        torch.cuda.synchronize(); t0 = time.perf_counter(); case["step"](); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print(T, " ".join(f"{t:.1f}" for t in ts), flush=True)
    del case
