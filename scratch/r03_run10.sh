#!/bin/bash
# final build (wgrad workspace combine opt-in, default atomics): GPU suite, default bench line, and the suite once more with MAEST_TN_REDUCE=1
export TMPDIR=/tmp
mkdir -p gpurun_out/r03j
timeout 1800 python -m pytest tests -m gpu -q -x > gpurun_out/r03j/pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/r03j/pytest.log
tail -4 gpurun_out/r03j/pytest.log
timeout 600 python bench.py > gpurun_out/r03j/bench_default.json 2> gpurun_out/r03j/bench_default.err; tail -c 300 gpurun_out/r03j/bench_default.json
MAEST_TN_REDUCE=1 timeout 1800 python -m pytest tests -m gpu -q -x -k "model or module or train or fullsize or dist or kernels" > gpurun_out/r03j/pytest_tn1.log 2>&1; echo "pytest exit $?" >> gpurun_out/r03j/pytest_tn1.log
tail -4 gpurun_out/r03j/pytest_tn1.log
