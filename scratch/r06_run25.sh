#!/bin/bash
mkdir -p gpurun_out/r06_dp1
run() { tag=$1; shift; env "$@" python bench.py --no-cpu-baseline --no-kernel-timing --no-side-cases --steps 20 $EXTRA > gpurun_out/r06_dp1/$tag.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/r06_dp1/$tag.json').read().strip().splitlines()[-1]); print('$tag', d['ms_per_step'])"; }
for rep in 1 2; do
EXTRA="" run plain_$rep A=1
EXTRA="--force-collective" run forced_timed_$rep A=1
EXTRA="--force-collective" run forced_untimed_$rep MAEST_DP_BUCKET_TIMING=0
done
