#!/bin/bash
# build (locally) a second library with MAEST_ATTN_PROF (clock stamps inside attn_bwd_fused2_kernel); run scratch/attn_prof.py on the GPU box
cd $(dirname $0)/..
mkdir -p maest_amd/build_prof
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -DMAEST_ATTN_PROF $EXTRA \
  -c maest_amd/csrc/attention.hip -o maest_amd/build_prof/attention.hip.o || exit 1
objs=$(ls maest_amd/build/*.o | grep -v attention)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o maest_amd/libmaest_hip_prof.so $objs maest_amd/build_prof/attention.hip.o
ls -la maest_amd/libmaest_hip_prof.so
