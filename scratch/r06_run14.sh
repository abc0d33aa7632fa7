#!/bin/bash
# final tree: whole GPU suite, smoke, default bench line
mkdir -p gpurun_out/r06q
python -m pytest tests -q -m gpu -x 2>&1 | tail -8 > gpurun_out/r06q/gpu_pytest_tail.txt
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/r06q/smoke.txt 2>&1; echo "smoke rc=$?" >> gpurun_out/r06q/smoke.txt
python bench.py > gpurun_out/r06q/bench_default_line.json 2> gpurun_out/r06q/bench_err.txt
tail -3 gpurun_out/r06q/gpu_pytest_tail.txt; tail -2 gpurun_out/r06q/smoke.txt; cut -c1-300 gpurun_out/r06q/bench_default_line.json
