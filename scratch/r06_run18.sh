#!/bin/bash
mkdir -p gpurun_out/r06g
MAEST_EVAL_STREAMS=2 python bench.py --no-cpu-baseline > gpurun_out/r06g/line_streams2.json 2> gpurun_out/r06g/err2.txt
MAEST_EVAL_STREAMS=1 python bench.py --no-cpu-baseline > gpurun_out/r06g/line_streams1.json 2> gpurun_out/r06g/err1.txt
MAEST_EVAL_STREAMS=2 python bench.py --no-cpu-baseline > gpurun_out/r06g/line_streams2b.json 2> gpurun_out/r06g/err2b.txt
python - <<'PY'
import json
for f in ("line_streams2", "line_streams1", "line_streams2b"):
    d = json.loads(open(f"gpurun_out/r06g/{f}.json").read().strip().splitlines()[-1])
    print(f, d["value"], {k: (d[k]["value"], d[k]["brackets_ms_per_step"], d[k].get("device_alloc_free_retry")) for k in ("infer", "infer_parity", "infer_fp16", "ts", "train30s")})
PY
