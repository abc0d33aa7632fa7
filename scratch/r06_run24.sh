#!/bin/bash
# what the data-parallel code path costs on one rank (forced one-rank RCCL communicator), and which part of it
mkdir -p gpurun_out/r06_dp1
run() { tag=$1; shift; env "$@" python bench.py --no-cpu-baseline --no-kernel-timing --no-side-cases --steps 20 $EXTRA > gpurun_out/r06_dp1/$tag.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/r06_dp1/$tag.json').read().strip().splitlines()[-1]); print('$tag', d['ms_per_step'])"; }
for rep in 1 2; do
EXTRA="" run plain_$rep A=1
EXTRA="--force-collective" run forced_$rep A=1
EXTRA="--force-collective" run forced_persistent_$rep MAEST_GEMM_WGS=256 MAEST_GEMM_TAIL=0
EXTRA="" run plain_nonpersistent_$rep MAEST_GEMM_WGS=0 MAEST_GEMM_TAIL=1 MAEST_PERSISTENT_GEMM=0
done
