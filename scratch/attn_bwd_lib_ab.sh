#!/bin/bash
# two-kernel attention backward: library variants (MAEST_HIP_LIB), interleaved processes
for r in 1 2; do for l in base u3 u2; do echo "== lib $l"; MAEST_HIP_LIB=$PWD/maest_amd/libmaest_$l.so python scratch/attn_bwd_forms.py 2 2>&1 | grep -v amdgpu | cut -c1-75; done; done
