#!/bin/bash
# second form of tn256_reduce_kernel (no LDS, four splits in flight, one wave per 32 x 32 block): kernel-level and step-level A/B, TN tests
export TMPDIR=/tmp
mkdir -p gpurun_out/r03m
timeout 600 python scratch/tn_reduce_ab.py > gpurun_out/r03m/tn_reduce_ab.txt 2>&1; grep "^tokens" gpurun_out/r03m/tn_reduce_ab.txt
timeout 600 python -m pytest tests -m gpu -q -x -k "gemm_tn" > gpurun_out/r03m/pytest_tn.log 2>&1; tail -2 gpurun_out/r03m/pytest_tn.log
for i in 1 2 3; do
  for v in 0 1; do
    MAEST_TN_REDUCE=$v python bench.py --no-cpu-baseline --no-kernel-timing --no-side-cases --steps 20 2>/dev/null | grep '^{"metric"' > gpurun_out/r03m/train_${v}_$i.json
  done
done
for i in 1 2; do
  for v in 0 1; do
    MAEST_TN_REDUCE=$v python bench.py --frames 1876 --batch 128 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-side-cases 2>/dev/null | grep '^{"metric"' > gpurun_out/r03m/t30_${v}_$i.json
  done
done
python - <<'PY' | tee gpurun_out/r03m/ab_step.txt
import json
for tag, name, n in (("train", "training step configs[2]", 3), ("t30", "30 s training step (B = 128, N = 875)", 2)):
    print(name)
    for i in range(1, n + 1):
        a = json.load(open(f"gpurun_out/r03m/{tag}_0_{i}.json")); b = json.load(open(f"gpurun_out/r03m/{tag}_1_{i}.json"))
        print(f"  atomics {a['ms_per_step']:8.3f} ms   workspace {b['ms_per_step']:8.3f} ms   {b['ms_per_step'] - a['ms_per_step']:+7.3f} ms ({100 * (b['ms_per_step'] / a['ms_per_step'] - 1):+5.2f} %)")
PY
