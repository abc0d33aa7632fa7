#!/bin/bash
mkdir -p gpurun_out/r06h
python bench.py --no-cpu-baseline > gpurun_out/r06h/line_a.json 2> gpurun_out/r06h/err_a.txt
MAEST_EVAL_STREAMS=1 python bench.py --no-cpu-baseline > gpurun_out/r06h/line_b.json 2> gpurun_out/r06h/err_b.txt
python - <<'PY'
import json
for f in ("line_a", "line_b"):
    d = json.loads(open(f"gpurun_out/r06h/{f}.json").read().strip().splitlines()[-1])
    print(f, d["value"], {k: (d[k]["value"], d[k]["brackets_ms_per_step"]) for k in ("infer", "infer_parity", "infer_fp16", "ts", "train30s")})
PY
