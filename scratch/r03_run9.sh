#!/bin/bash
# wgrad split-K combine through a workspace: kernel-level A/B, the whole GPU suite, step-level pairs (MAEST_TN_REDUCE=0 against the default)
export TMPDIR=/tmp
mkdir -p gpurun_out/r03i
timeout 900 python scratch/tn_reduce_ab.py > gpurun_out/r03i/tn_reduce_ab.txt 2>&1; echo "exit $?" >> gpurun_out/r03i/tn_reduce_ab.txt
grep -v amdgpu.ids gpurun_out/r03i/tn_reduce_ab.txt
timeout 1800 python -m pytest tests -m gpu -q -x > gpurun_out/r03i/pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/r03i/pytest.log
tail -5 gpurun_out/r03i/pytest.log
for i in 1 2 3; do
  for v in 0 1; do
    MAEST_TN_REDUCE=$v python bench.py --no-cpu-baseline --no-kernel-timing --no-side-cases --steps 20 2>/dev/null | grep '^{"metric"' > gpurun_out/r03i/train_${v}_$i.json
    MAEST_TN_REDUCE=$v python bench.py --frames 1876 --batch 128 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-side-cases 2>/dev/null | grep '^{"metric"' > gpurun_out/r03i/t30_${v}_$i.json
  done
done
python - <<'PY' | tee gpurun_out/r03i/ab_step.txt
import json
for tag, name in (("train", "training step configs[2]"), ("t30", "30 s training step (B = 128, N = 875)")):
    print(name)
    for i in (1, 2, 3):
        a = json.load(open(f"gpurun_out/r03i/{tag}_0_{i}.json")); b = json.load(open(f"gpurun_out/r03i/{tag}_1_{i}.json"))
        print(f"  atomics {a['ms_per_step']:8.3f} ms   workspace {b['ms_per_step']:8.3f} ms   {b['ms_per_step'] - a['ms_per_step']:+7.3f} ms ({100 * (b['ms_per_step'] / a['ms_per_step'] - 1):+5.2f} %)")
PY
