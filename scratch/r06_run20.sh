#!/bin/bash
# weight-gradient launch width beside the 16 x 16 NT kernel (round 5 tuned 128 against the 32 x 32 kernel)
mkdir -p gpurun_out/r06_wgw
for rep in 1 2; do
for w in 128 96 160 192 224 0; do
  MAEST_WGRAD_WGS=$w python bench.py --no-cpu-baseline --no-kernel-timing --no-side-cases --steps 20 > gpurun_out/r06_wgw/w${w}_$rep.json 2>/dev/null
done; done
python - <<'PY'
import json
for w in (96, 128, 160, 192, 224, 0):
    v = [json.loads(open(f"gpurun_out/r06_wgw/w{w}_{r}.json").read().strip().splitlines()[-1])["ms_per_step"] for r in (1, 2)]
    print("MAEST_WGRAD_WGS=%3d: %.3f / %.3f ms per step" % (w, v[0], v[1]))
PY
