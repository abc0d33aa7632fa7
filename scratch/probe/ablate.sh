#!/bin/bash
# build (locally) or run (on the GPU box) the ablation variants of the 256x256 NT GEMM main loop
cd $(dirname $0)
if [ "$1" = build ]; then
  for v in FULL NO_DMA; do
    d=""; [ $v != FULL ] && d="-DMAEST_ABLATE_$v"
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-result $d \
      ../../maest_amd/csrc/gemm256.hip ../../maest_amd/csrc/capi.hip ablate_main.cpp -o ablate_$v &
  done; wait
else
  for v in FULL NO_DMA; do ./ablate_$v wide_$v; done
fi
