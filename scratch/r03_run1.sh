#!/bin/bash
# round 3, first GPU call: the whole -m gpu suite on the new code, the per-shape / per-epilogue GEMM table, the default bench line
mkdir -p gpurun_out/r03a
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03a/pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/r03a/pytest.log
tail -5 gpurun_out/r03a/pytest.log
timeout 300 python scratch/gemm_ab.py > gpurun_out/r03a/gemm_ab.txt 2>&1
cat gpurun_out/r03a/gemm_ab.txt
timeout 600 python bench.py > gpurun_out/r03a/bench.json 2> gpurun_out/r03a/bench.err; tail -c 6000 gpurun_out/r03a/bench.json
timeout 300 python bench.py --force-collective --no-cpu-baseline --no-kernel-timing > gpurun_out/r03a/bench_fc.json 2> gpurun_out/r03a/bench_fc.err; cat gpurun_out/r03a/bench_fc.json; tail -3 gpurun_out/r03a/bench_fc.err
