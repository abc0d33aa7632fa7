#!/bin/bash
# committed head: GPU suite (default) and again with the deterministic wgrad combine, default bench line
export TMPDIR=/tmp
mkdir -p gpurun_out/r03n
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r03n/pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/r03n/pytest.log; tail -3 gpurun_out/r03n/pytest.log
MAEST_TN_REDUCE=1 timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r03n/pytest_tn1.log 2>&1; echo "pytest exit $?" >> gpurun_out/r03n/pytest_tn1.log; tail -3 gpurun_out/r03n/pytest_tn1.log
timeout 600 python bench.py > gpurun_out/r03n/bench_default.json 2> gpurun_out/r03n/bench_default.err; tail -c 200 gpurun_out/r03n/bench_default.json
