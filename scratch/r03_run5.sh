#!/bin/bash
# attention forward: waves per workgroup sweep; then the whole GPU suite and the default bench line on this build
export TMPDIR=/tmp
mkdir -p gpurun_out/r03e
timeout 600 python scratch/attn_fwd_waves.py > gpurun_out/r03e/attn_fwd_waves.txt 2>&1; echo "exit $?" >> gpurun_out/r03e/attn_fwd_waves.txt
cat gpurun_out/r03e/attn_fwd_waves.txt
timeout 1800 python -m pytest tests -m gpu -q -x > gpurun_out/r03e/pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/r03e/pytest.log
tail -6 gpurun_out/r03e/pytest.log
timeout 600 python bench.py > gpurun_out/r03e/bench_default.json 2> gpurun_out/r03e/bench_default.err; tail -c 600 gpurun_out/r03e/bench_default.json
