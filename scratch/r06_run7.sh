#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r06g; mkdir -p $O
timeout 600 python -m pytest tests -x -q -m gpu -k "fp16" > $O/pytest_fp16.txt 2>&1; echo "pytest exit $?" >> $O/pytest_fp16.txt
