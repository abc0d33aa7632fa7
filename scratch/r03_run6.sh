#!/bin/bash
# NT main loop: wave-priority variants (MAEST_NT_PRIO 0..3), interleaved twice; then the fuzz / soak scripts on this build
export TMPDIR=/tmp
mkdir -p gpurun_out/r03f
cd scratch/probe
for rep in 1 2; do for v in 0 1 2 3; do timeout 120 ./ablw_PRIO$v PRIO$v; done; done > ../../gpurun_out/r03f/nt_prio.txt 2>&1
cd ../..
cat gpurun_out/r03f/nt_prio.txt | grep -E "M= 74240"
for s in fuzz_shapes fuzz_train soak; do timeout 600 python scratch/$s.py > gpurun_out/r03f/$s.txt 2>&1; echo "$s exit $?"; tail -3 gpurun_out/r03f/$s.txt; done
