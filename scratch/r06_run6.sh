#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r06f; mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu -k "fp16 or g1_eval" > $O/pytest_fp16.txt 2>&1; echo "pytest exit $?" >> $O/pytest_fp16.txt
timeout 600 python bench.py --mode infer --precision fp16 --steps 20 --no-cpu-baseline --no-side-cases > $O/bench_infer_fp16.json 2> $O/bench_infer_fp16.err
timeout 600 python bench.py --mode infer --steps 20 --no-cpu-baseline --no-side-cases > $O/bench_infer_bf16.json 2> $O/bench_infer_bf16.err
timeout 600 python bench.py --mode infer --precision fp16 --frames 1876 --batch 64 --steps 10 --no-cpu-baseline --no-side-cases > $O/bench_infer30s_fp16.json 2> $O/bench_infer30s_fp16.err
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_full.txt 2>&1; echo "pytest exit $?" >> $O/pytest_full.txt
