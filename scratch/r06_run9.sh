#!/bin/bash
# one box: the GEMM / bf16x3 tests after re-enabling the fp32-output forms, the fuzz, then the default line
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
TAG=${1:-r06b}; O=gpurun_out/$TAG; mkdir -p $O
if [ "$2" == "tests" ]; then
  timeout 900 python -m pytest tests -x -q -m gpu -k "gemm or bf16x3 or split or g1_eval or fp16" > $O/pytest_sel.txt 2>&1; echo "pytest exit $?" >> $O/pytest_sel.txt
  timeout 600 python scratch/fuzz_gemm_ow.py > $O/fuzz_gemm.txt 2>&1; echo "fuzz exit $?" >> $O/fuzz_gemm.txt
fi
( s=$(date +%s); timeout 900 python bench.py > $O/${TAG}_bench_default_line.json 2> $O/bench_default.err; echo "default line wall $(( $(date +%s) - s )) s" > $O/${TAG}_bench_default_wall.txt )
