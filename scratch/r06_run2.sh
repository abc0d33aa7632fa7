#!/bin/bash
# round 6, second GPU call: the deferred-store GEMM kernel -- parity, stand-alone A/B, step-level A/B; the MFMA-shape power probe
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r06b; mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu -k "deferred or one_wave_per_simd or test_gemm" > $O/pytest_gemm.txt 2>&1; echo "pytest exit $?" >> $O/pytest_gemm.txt
timeout 300 scratch/probe/mfma_power > $O/mfma_power.txt 2>&1
timeout 300 scratch/probe/mfma_power 1 >> $O/mfma_power.txt 2>&1
timeout 900 python scratch/r06_defer_ab.py > $O/defer_ab.txt 2>&1
timeout 900 bash scratch/ab_env.sh r06b_defer_train "MAEST_GEMM_DEFER=0" "MAEST_GEMM_DEFER=1" 3 > $O/ab_defer_train.txt 2>&1
timeout 900 bash scratch/ab_env.sh r06b_defer_infer "MAEST_GEMM_DEFER=0" "MAEST_GEMM_DEFER=1" 3 "--mode infer" > $O/ab_defer_infer.txt 2>&1
timeout 600 python bench.py --mode infer --steps 20 --no-cpu-baseline --no-side-cases > $O/bench_infer.json 2> $O/bench_infer.err
