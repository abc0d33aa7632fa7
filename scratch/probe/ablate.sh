#!/bin/bash
# build (locally) or run (on the GPU box) ablation variants of the gemm256.hip main loops (MAEST_ABLATE_* hooks)
cd $(dirname $0)
VARS="FULL NO_EPI"
if [ "$1" = build ]; then
  for v in $VARS; do
    d=""
    case $v in
      NO_DMA_NO_DSREAD) d="-DMAEST_ABLATE_NO_DMA -DMAEST_ABLATE_NO_DSREAD";;
      FULL) d="";;
      *) d="-DMAEST_ABLATE_$v";;
    esac
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-result $d \
      ../../maest_amd/csrc/gemm256.hip ../../maest_amd/csrc/capi.hip ablate_main.cpp -o ablate_$v &
  done; wait
else
  for v in $VARS; do ./ablate_$v $v ${1:-}; done
fi
