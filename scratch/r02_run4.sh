#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r02d; mkdir -p $O
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "attention" > $O/pytest_attn.log 2>&1; tail -3 $O/pytest_attn.log
timeout 300 python scratch/attn_ablate.py > $O/attn_ablate.txt 2>&1; cat $O/attn_ablate.txt
timeout 300 python scratch/gemm_epi_ablate.py > $O/gemm_epi_ablate.txt 2>&1; cat $O/gemm_epi_ablate.txt
