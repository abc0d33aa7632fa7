#!/bin/bash
# A/B on ONE box: the last block on the head's tokens (default) against the complete last block, training step and
# inference, alternating, two rounds -> gpurun_out/$1/ab_last_block.txt
O=gpurun_out/${1:-r02c}; mkdir -p $O
for r in 1 2; do
  for mode in train infer; do
    for flag in "" "--complete-last-block"; do
      python bench.py --mode $mode --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing $flag 2>/dev/null | tail -1 | \
        python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$mode', '${flag:-head-tokens}', d['value'], 'clips/s', d['ms_per_step'], 'ms', 'executed', d['executed_flop_fraction'])"
    done
  done
done | tee $O/ab_last_block.txt
