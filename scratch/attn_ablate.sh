#!/bin/bash
# timing ablations of the persistent attention backward: `build` (locally) makes one library per MAEST_ABLATE_F3 mask, `run` (GPU box) times them
cd $(dirname $0)/..
MASKS=${MASKS:-"0 1 2 4 6 8 16 24 32 64 7 127"}
if [ "$1" = build ]; then
  mkdir -p maest_amd/build_abl
  objs=$(ls maest_amd/build/*.o | grep -v attention)
  for m in $MASKS; do
    ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -DMAEST_ABLATE_F3=$m \
        -c maest_amd/csrc/attention.hip -o maest_amd/build_abl/attention_$m.o && \
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o maest_amd/build_abl/libabl_$m.so $objs maest_amd/build_abl/attention_$m.o ) &
  done; wait; ls maest_amd/build_abl/*.so
else
  for m in $MASKS; do echo "mask $m: $(python scratch/attn_bench.py maest_amd/build_abl/libabl_$m.so 2>&1 | grep 'N=290')"; done
fi
