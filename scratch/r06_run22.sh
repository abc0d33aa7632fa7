#!/bin/bash
# dgrad launch width beside the 128-wide wgrads (16 x 16 NT kernel)
mkdir -p gpurun_out/r06_bgw
for rep in 1 2; do
for w in 256 240 224 208; do
  MAEST_BWD_GEMM_WGS=$w python bench.py --no-cpu-baseline --no-kernel-timing --no-side-cases --steps 20 > gpurun_out/r06_bgw/w${w}_$rep.json 2>/dev/null
done; done
python - <<'PY'
import json
for w in (256, 240, 224, 208):
    v = [json.loads(open(f"gpurun_out/r06_bgw/w{w}_{r}.json").read().strip().splitlines()[-1])["ms_per_step"] for r in (1, 2)]
    print("MAEST_BWD_GEMM_WGS=%3d: %.3f / %.3f ms per step" % (w, v[0], v[1]))
PY
