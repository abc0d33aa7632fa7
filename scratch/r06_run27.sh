#!/bin/bash
mkdir -p gpurun_out/r06_dp1
run() { tag=$1; shift; env "$@" python bench.py --no-cpu-baseline --no-kernel-timing --no-side-cases --steps 20 $EXTRA > gpurun_out/r06_dp1/$tag.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/r06_dp1/$tag.json').read().strip().splitlines()[-1]); print('$tag', d['ms_per_step'])"; }
EXTRA="" run plain_default A=1
EXTRA="--force-collective" run forced_default A=1
EXTRA="--force-collective" run forced_hwq4 GPU_MAX_HW_QUEUES=4
EXTRA="--mode infer" run infer_default A=1
EXTRA="--mode infer" run infer_hwq4 GPU_MAX_HW_QUEUES=4
GPU_MAX_HW_QUEUES=8 python scratch/r06_stream_identity.py 2>&1 | tail -6
