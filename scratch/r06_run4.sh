#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r06d; mkdir -p $O
timeout 1200 python -m pytest tests -x -q -m gpu -k "gemm" > $O/pytest_gemm.txt 2>&1; echo "pytest exit $?" >> $O/pytest_gemm.txt
timeout 900 python scratch/r06_mfma16_ab.py > $O/mfma16_ab.txt 2>&1
timeout 600 python bench.py --steps 20 --no-cpu-baseline --no-side-cases > $O/bench_train.json 2> $O/bench_train.err
timeout 600 python bench.py --mode infer --steps 20 --no-cpu-baseline --no-side-cases > $O/bench_infer.json 2> $O/bench_infer.err
