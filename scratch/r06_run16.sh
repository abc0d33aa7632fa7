#!/bin/bash
# balanced persistent grid (as few workgroups as finish in the same number of rounds) vs 256
bash scratch/ab_env.sh r06_balance_train "MAEST_GEMM_WGS=256 MAEST_GEMM_TAIL=0" "MAEST_GEMM_WGS=-256 MAEST_GEMM_TAIL=0" 3 > gpurun_out/r06_balance_train.txt 2>&1
bash scratch/ab_env.sh r06_balance_infer1 "MAEST_GEMM_WGS=256 MAEST_GEMM_TAIL=0 MAEST_EVAL_STREAMS=1" "MAEST_GEMM_WGS=-256 MAEST_GEMM_TAIL=0 MAEST_EVAL_STREAMS=1" 2 "--mode infer" > gpurun_out/r06_balance_infer1.txt 2>&1
bash scratch/ab_env.sh r06_balance_infer2 "MAEST_GEMM_WGS=256 MAEST_GEMM_TAIL=0" "MAEST_GEMM_WGS=-256 MAEST_GEMM_TAIL=0" 2 "--mode infer" > gpurun_out/r06_balance_infer2.txt 2>&1
cat gpurun_out/r06_balance_train.txt gpurun_out/r06_balance_infer1.txt gpurun_out/r06_balance_infer2.txt
