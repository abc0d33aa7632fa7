#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r06h; mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu -k "deferred" > $O/pytest_defer.txt 2>&1; echo "pytest exit $?" >> $O/pytest_defer.txt
timeout 900 python scratch/r06_defer2_ab.py > $O/defer2_ab.txt 2>&1
timeout 900 bash scratch/ab_env.sh r06h_defer_infer "MAEST_GEMM_DEFER=0" "MAEST_GEMM_DEFER=12" 3 "--mode infer" > $O/ab_defer_infer.txt 2>&1
timeout 900 bash scratch/ab_env.sh r06h_defer_train "MAEST_GEMM_DEFER=0" "MAEST_GEMM_DEFER=12" 3 > $O/ab_defer_train.txt 2>&1
