#!/bin/bash
# end-of-round evidence on the final build: whole GPU suite, then the full profile round (r03c)
export TMPDIR=/tmp
mkdir -p gpurun_out/r03h
timeout 1800 python -m pytest tests -m gpu -q -x > gpurun_out/r03h/pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/r03h/pytest.log
tail -4 gpurun_out/r03h/pytest.log
timeout 2400 bash scratch/profile_round.sh r03c > gpurun_out/r03h/profile_round.log 2>&1; echo "profile exit $?"
ls gpurun_out/r03c | head -40
timeout 600 python bench.py > gpurun_out/r03h/bench_default.json 2> gpurun_out/r03h/bench_default.err; tail -c 400 gpurun_out/r03h/bench_default.json
