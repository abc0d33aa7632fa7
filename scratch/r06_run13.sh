#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r06k; mkdir -p $O
timeout 600 python scratch/r06_attn_crude16.py > $O/attn_crude16.txt 2>&1
