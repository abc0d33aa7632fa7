#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r06c; mkdir -p $O
timeout 1200 python scratch/owd_ablate_run.py 3 > $O/owd_ablate.txt 2>&1
