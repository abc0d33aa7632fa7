#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r02c; mkdir -p $O
timeout 300 python scratch/attn_ablate.py > $O/attn_ablate.txt 2>&1; cat $O/attn_ablate.txt
timeout 600 python scratch/stagger_sweep.py > $O/stagger.txt 2>&1; cat $O/stagger.txt
timeout 900 bash scratch/pmc_util.sh r02c > $O/pmc_util.txt 2>&1; tail -40 $O/pmc_util.txt
