#!/bin/bash
# A/B of the wgrad launch width in the overlapped training step (the wgrad runs on the side stream beside the dgrad chain: a narrower launch
# -- fewer split-K atomics, longer K per workgroup -- leaves CUs to the main stream's kernel)
mkdir -p gpurun_out/r05t
for v in "$@"; do
  r=$(env $v python bench.py --no-cpu-baseline --no-kernel-timing --no-side-cases --steps 20 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")
  echo "$v : $r"
done
