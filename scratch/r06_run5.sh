#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r06e; mkdir -p $O
timeout 900 python scratch/r06_mfma16_ab.py > $O/mfma16_ab.txt 2>&1
timeout 900 bash scratch/ab_env.sh r06e_mfma_train "MAEST_HIP_LIB=$R/scratch/pw_abl/libmaest_o-mfma32.so" "MAEST_X=1" 3 > $O/ab_mfma_train.txt 2>&1
timeout 900 bash scratch/ab_env.sh r06e_mfma_infer "MAEST_HIP_LIB=$R/scratch/pw_abl/libmaest_o-mfma32.so" "MAEST_X=1" 3 "--mode infer" > $O/ab_mfma_infer.txt 2>&1
