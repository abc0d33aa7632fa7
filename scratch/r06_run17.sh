#!/bin/bash
mkdir -p gpurun_out/r06f
python -m pytest tests/test_fullsize_gpu.py tests/test_model_gpu.py -q -m gpu -x 2>&1 | tail -5 > gpurun_out/r06f/pytest_tail.txt
python bench.py > gpurun_out/r06f/bench_default_line.json 2> gpurun_out/r06f/bench_err.txt
tail -3 gpurun_out/r06f/pytest_tail.txt; tail -3 gpurun_out/r06f/bench_err.txt; cut -c1-200 gpurun_out/r06f/bench_default_line.json
