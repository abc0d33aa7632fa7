#!/bin/bash
mkdir -p gpurun_out/r03b
timeout 300 python scratch/gemm_tail_ab.py > gpurun_out/r03b/gemm_tail_ab.txt 2>&1
cat gpurun_out/r03b/gemm_tail_ab.txt
timeout 300 python scratch/gemm_ab.py > gpurun_out/r03b/gemm_ab.txt 2>&1
grep -E "M =|mul|resid|gelu\+aux" gpurun_out/r03b/gemm_ab.txt
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r03b/pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/r03b/pytest.log
tail -8 gpurun_out/r03b/pytest.log
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r03b/bench.json 2> gpurun_out/r03b/bench.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03b/bench.json").read().strip().splitlines()[-1])
print("train", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["kernel_ms_per_step"])
print("infer", d["infer"]["value"], d["infer"]["roofline"]["frac"]); print("train30s", d["train30s"]["value"], d["train30s"]["roofline"]["frac"])
PY
