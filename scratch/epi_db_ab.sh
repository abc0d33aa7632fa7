#!/bin/bash
# double-buffered C staging of the bf16-output epilogues (OW_EPI_DB) against the one-buffer build: kernel tests, per-form GEMM times, the step
python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "gemm" 2>&1 | tail -2
for l in base hip base hip; do
  echo "== lib $l"; MAEST_HIP_LIB=$PWD/maest_amd/libmaest_$l.so python scratch/gemm_ow_forms.py 2>&1 | grep -v amdgpu | cut -c1-62
done
for r in 1 2 3; do
for l in base hip; do
  r=$(MAEST_HIP_LIB=$PWD/maest_amd/libmaest_$l.so python bench.py --no-cpu-baseline --no-kernel-timing --no-side-cases --steps 20 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")
  echo "step $l : $r"
done; done
for l in base hip base hip; do
  r=$(MAEST_HIP_LIB=$PWD/maest_amd/libmaest_$l.so python bench.py --mode infer --no-cpu-baseline --no-kernel-timing --no-side-cases --steps 20 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")
  echo "infer $l : $r"
done
