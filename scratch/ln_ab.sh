#!/bin/bash
# LayerNorm backward variants as separate builds (MAEST_HIP_LIB): stand-alone kernel time and the training step, alternating
for r in 1 2; do
for l in base ln3 ln2; do
  echo -n "$l: "; MAEST_HIP_LIB=$PWD/maest_amd/libmaest_$l.so python scratch/ln_bench.py 2>&1 | grep layernorm_bwd
done; done
for r in 1 2; do
for l in base ln3 ln2; do
  r=$(MAEST_HIP_LIB=$PWD/maest_amd/libmaest_$l.so python bench.py --no-cpu-baseline --no-kernel-timing --no-side-cases --steps 20 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")
  echo "step $l : $r"
done; done
