#!/bin/bash
for v in "$@"; do
  r=$(env $v python bench.py --no-cpu-baseline --no-kernel-timing --no-side-cases --steps 10 --warmup 3 --frames 1876 --batch 128 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")
  echo "train30s $v : $r"
done
for v in "$@"; do
  r=$(env $v python bench.py --no-cpu-baseline --no-kernel-timing --no-side-cases --steps 10 --warmup 3 --mode ts 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")
  echo "ts $v : $r"
done
