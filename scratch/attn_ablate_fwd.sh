#!/bin/bash
# timing ablations of the attention forward: `build` (locally) one library per MAEST_ABLATE_FWD mask, `run` (GPU box) times them
cd $(dirname $0)/..
MASKS=${MASKS:-"0 1 2 4 8 16 3 7 15 31"}
if [ "$1" = build ]; then
  mkdir -p maest_amd/build_abl
  objs=$(ls maest_amd/build/*.o | grep -v attention)
  for m in $MASKS; do
    ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -DMAEST_ABLATE_FWD=$m \
        -c maest_amd/csrc/attention.hip -o maest_amd/build_abl/attention_f$m.o && \
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o maest_amd/build_abl/libablf_$m.so $objs maest_amd/build_abl/attention_f$m.o ) &
  done; wait; ls maest_amd/build_abl/libablf_*.so | wc -l
else
  for m in $MASKS; do echo "mask $m:"; python scratch/attn_bench.py maest_amd/build_abl/libablf_$m.so 2>&1 | grep 'N=' | cut -c1-48; done
fi
