#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r06i; mkdir -p $O
for b in 256 128 96 64; do
  timeout 300 python bench.py --mode infer --batch $b --steps 30 --no-cpu-baseline --no-side-cases 2>/dev/null | tail -1 > $O/infer_b$b.json
done
for b in 256 128; do
  timeout 300 python bench.py --batch $b --steps 20 --no-cpu-baseline --no-side-cases 2>/dev/null | tail -1 > $O/train_b$b.json
done
