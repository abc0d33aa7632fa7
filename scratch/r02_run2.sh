#!/bin/bash
# benches one by one (each bounded), then breakdown / probes
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r02b; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -x -q -k "attention or hip_graph or g5 or weight_averager" > $O/pytest_sel.log 2>&1; tail -5 $O/pytest_sel.log
timeout 300 python scratch/attn_bench.py > $O/attn_bench.txt 2>&1; cat $O/attn_bench.txt
timeout 300 python bench.py --steps 10 --warmup 3 2>$O/bench_train.err | tail -1 > $O/bench_train.json; cat $O/bench_train.json; tail -3 $O/bench_train.err
timeout 300 python bench.py --mode infer --steps 10 --warmup 3 2>$O/bench_infer.err | tail -1 > $O/bench_infer.json; cat $O/bench_infer.json
timeout 200 scratch/probe/mfma_rate > $O/mfma_rate.txt 2>&1; cat $O/mfma_rate.txt
timeout 300 python scratch/step_breakdown.py > $O/step_breakdown.txt 2>&1; tail -45 $O/step_breakdown.txt
