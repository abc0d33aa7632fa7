#!/bin/bash
# round 6, first GPU call: new tests, column-panel A/B (stand-alone + counters + step level), the vendor yardstick with clocks, default line
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r06a; mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu -k "column_panels or qkv_bias or wgrad_launch_width or one_wave_per_simd or bucket or rccl_one_rank or gpus_8" > $O/pytest_new.txt 2>&1; echo "pytest exit $?" >> $O/pytest_new.txt
timeout 600 python scratch/r06_panel_ab.py > $O/panel_ab.txt 2>&1
timeout 900 bash scratch/r06_pmc_gemm.sh > $O/pmc_gemm.txt 2>&1
timeout 900 bash scratch/ab_env.sh r06a_panel_train "MAEST_GEMM_PANEL=0" "MAEST_GEMM_PANEL=-1" 3 > $O/ab_panel_train.txt 2>&1
timeout 900 bash scratch/ab_env.sh r06a_panel_infer "MAEST_GEMM_PANEL=0" "MAEST_GEMM_PANEL=-1" 3 "--mode infer" > $O/ab_panel_infer.txt 2>&1
( s=$(date +%s); timeout 900 python bench.py > $O/bench_default_line.json 2> $O/bench_default.err; echo "default line wall $(( $(date +%s) - s )) s" > $O/bench_default_wall.txt )
timeout 1500 bash scratch/r06_lib_vs_own.sh > $O/lib_vs_own.txt 2>&1
