#!/bin/bash
mkdir -p gpurun_out/r06_dp1
run() { tag=$1; shift; env "$@" python bench.py --no-cpu-baseline --no-kernel-timing --no-side-cases --steps 20 $EXTRA > gpurun_out/r06_dp1/$tag.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/r06_dp1/$tag.json').read().strip().splitlines()[-1]); print('$tag', d['ms_per_step'])"; }
EXTRA="" run plain A=1
EXTRA="--force-collective" run forced A=1
EXTRA="--force-collective" run forced_hwq8 GPU_MAX_HW_QUEUES=8
EXTRA="--force-collective" run forced_hwq16 GPU_MAX_HW_QUEUES=16
EXTRA="" run plain_hwq8 GPU_MAX_HW_QUEUES=8
EXTRA="--force-collective" run forced_nccl_hp TORCH_NCCL_HIGH_PRIORITY=1
