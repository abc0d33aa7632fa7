# logmel_kernel alone at the bench shapes (10 s and 30 s clips), HIP events; prints GB/s against the 8 TB/s HBM peak
import sys, torch
sys.path.insert(0, ".")
import bench
for clips, samples in ((256, 160000), (128, 480000), (1, 160000)):
    r = bench.time_mel_kernel("cuda", clips, samples, reps=20)
    print(clips, samples, r["avg_launch_ms"], "ms", r["achieved"], "GB/s  frac", r["frac"], " clips/s", r["clips_per_s"])
