#!/bin/bash
# the 128-row tail launch at the 30 s training shape (N = 768 GEMMs: 1314 tiles = 5.13 rounds) and at inference (1680 = 6.56; tail rule does not fire there)
bash scratch/ab_env.sh r06_tail30 "MAEST_GEMM_WGS=256 MAEST_GEMM_TAIL=0" "MAEST_GEMM_WGS=256 MAEST_GEMM_TAIL=1" 3 "--frames 1876 --batch 128 --patchout 90" > gpurun_out/r06_tail30.txt 2>&1
cat gpurun_out/r06_tail30.txt
