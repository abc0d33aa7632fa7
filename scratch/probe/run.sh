#!/bin/bash
# run a probe binary on the GPU box, output under gpurun_out/<dir>/<name>.txt:  scratch/probe/run.sh r02m nt4w
mkdir -p gpurun_out/$1
timeout ${3:-200} scratch/probe/$2 > gpurun_out/$1/$2.txt 2>&1
cat gpurun_out/$1/$2.txt
