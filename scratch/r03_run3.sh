#!/bin/bash
mkdir -p gpurun_out/r03c
timeout 1800 python -m pytest tests -m gpu -q -x > gpurun_out/r03c/pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/r03c/pytest.log
tail -6 gpurun_out/r03c/pytest.log
scratch/ab.sh r03c_ab_delta "" "--no-fold-delta" 3 | tee gpurun_out/r03c/ab_fold_delta.txt
timeout 300 python scratch/gemm_ab.py > gpurun_out/r03c/gemm_ab.txt 2>&1
grep -E "M =|fc1 " gpurun_out/r03c/gemm_ab.txt
