#!/bin/bash
# the default line of the final tree on this box
tag=${1:-x}
mkdir -p gpurun_out/r06i
python bench.py > gpurun_out/r06i/bench_default_line_$tag.json 2> gpurun_out/r06i/err_$tag.txt
cut -c1-160 gpurun_out/r06i/bench_default_line_$tag.json
