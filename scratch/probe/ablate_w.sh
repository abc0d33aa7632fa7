#!/bin/bash
# build (locally) or run (on the GPU box) ablation variants of the full-line NT kernel's main loop
cd $(dirname $0)
VARS=${VARS:-"FULL NO_DMA NO_MFMA NO_DSREAD NO_BARRIER NO_DMA+NO_DSREAD NO_MFMA+NO_DSREAD NO_DMA+NO_BARRIER NO_MFMA+NO_DMA"}
if [ "$1" = build ]; then
  for v in $VARS; do
    d=""
    if [ $v != FULL ]; then for part in ${v//+/ }; do d="$d -DMAEST_ABLATE_${part/=/=}"; done; fi
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-result $d \
      ../../maest_amd/csrc/gemm256.hip ../../maest_amd/csrc/capi.hip ablate_w.cpp -o ablw_$v &
  done; wait
else
  for v in $VARS; do ./ablw_$v $v; done
fi
