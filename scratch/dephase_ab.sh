#!/bin/bash
# persistent NT GEMM with the workgroups' tile loops staggered over P phases (MAEST_GEMM_DEPHASE = P, + 1024: also when every list has the same length)
for d in 0 4 1028 8 1032 0 2 1026; do
  echo "== MAEST_GEMM_DEPHASE=$d"; MAEST_GEMM_DEPHASE=$d MAEST_GEMM_WGS=256 python scratch/gemm_ow_forms.py 2>&1 | grep -v amdgpu | cut -c1-62
done
for d in 0 4 1028 0 4 1028 8; do
  r=$(MAEST_GEMM_DEPHASE=$d python bench.py --no-cpu-baseline --no-kernel-timing --no-side-cases --steps 20 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")
  echo "step dephase $d : $r"
done
for d in 0 4 1028 0 4 1028; do
  r=$(MAEST_GEMM_DEPHASE=$d python bench.py --mode infer --no-cpu-baseline --no-kernel-timing --no-side-cases --steps 20 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")
  echo "infer dephase $d : $r"
done
