#!/bin/bash
# attention store tails as 16-byte row pieces: kernel-level old / new interleaved, attention tests, then the bench step old / new (library file swapped)
export TMPDIR=/tmp
mkdir -p gpurun_out/r03g
timeout 900 python scratch/attn_store_ab.py scratch/probe/libmaest_hip_old.so > gpurun_out/r03g/attn_store_ab.txt 2>&1; echo "exit $?" >> gpurun_out/r03g/attn_store_ab.txt
cat gpurun_out/r03g/attn_store_ab.txt
timeout 1200 python -m pytest tests -m gpu -q -x -k "attention or attn or kernels" > gpurun_out/r03g/pytest_attn.log 2>&1; echo "pytest exit $?" >> gpurun_out/r03g/pytest_attn.log
tail -4 gpurun_out/r03g/pytest_attn.log
cp maest_amd/libmaest_hip.so /tmp/new.so
for i in 1 2 3; do
  for v in old new; do
    if [ $v = old ]; then cp scratch/probe/libmaest_hip_old.so maest_amd/libmaest_hip.so; else cp /tmp/new.so maest_amd/libmaest_hip.so; fi
    python bench.py --no-cpu-baseline --no-kernel-timing --no-side-cases --steps 20 2>/dev/null | grep '^{"metric"' > gpurun_out/r03g/train_${v}_$i.json
    python bench.py --mode infer --no-cpu-baseline --no-kernel-timing --no-side-cases --steps 20 2>/dev/null | grep '^{"metric"' > gpurun_out/r03g/infer_${v}_$i.json
    python bench.py --frames 1876 --batch 128 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-side-cases 2>/dev/null | grep '^{"metric"' > gpurun_out/r03g/t30_${v}_$i.json
  done
done
cp /tmp/new.so maest_amd/libmaest_hip.so
python - <<'PY' | tee gpurun_out/r03g/ab_step.txt
import json
for tag, name in (("train", "training step configs[2]"), ("infer", "inference configs[1]"), ("t30", "30 s training step (B = 128, N = 875)")):
    print(name)
    for i in (1, 2, 3):
        a = json.load(open(f"gpurun_out/r03g/{tag}_old_{i}.json")); b = json.load(open(f"gpurun_out/r03g/{tag}_new_{i}.json"))
        print(f"  old {a['ms_per_step']:8.3f} ms   new {b['ms_per_step']:8.3f} ms   {b['ms_per_step'] - a['ms_per_step']:+7.3f} ms ({100 * (b['ms_per_step'] / a['ms_per_step'] - 1):+5.2f} %)")
PY
