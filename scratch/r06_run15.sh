#!/bin/bash
mkdir -p gpurun_out/r06e
python -m pytest tests/test_model_gpu.py -q -m gpu -x -k "two_streams or hip_graph or inference or eval" 2>&1 | tail -4 > gpurun_out/r06e/pytest_two_streams.txt
python scratch/r06_eval_two_streams.py > gpurun_out/r06e/eval_two_streams.txt 2>&1
cat gpurun_out/r06e/pytest_two_streams.txt; cat gpurun_out/r06e/eval_two_streams.txt | tail -12
