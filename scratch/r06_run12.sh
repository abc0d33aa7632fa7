#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r06j; mkdir -p $O
timeout 600 python scratch/r06_attn_fwd_290.py > $O/attn_fwd_290.txt 2>&1
timeout 300 python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.txt 2>&1; echo "smoke exit $?" >> $O/smoke.txt
