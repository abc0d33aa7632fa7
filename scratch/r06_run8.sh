#!/bin/bash
# round 6 evidence set on one box: full GPU suite, default line (wall-clocked), the per-configuration JSONs + rocprofv3 stats + PMC passes
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r06a; mkdir -p $O
( s=$(date +%s); timeout 900 python bench.py > $O/r06a_bench_default_line.json 2> $O/bench_default.err; echo "default line wall $(( $(date +%s) - s )) s" > $O/r06a_bench_default_wall.txt )
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_full.txt 2>&1; echo "pytest exit $?" >> $O/pytest_full.txt
tail -4 $O/pytest_full.txt > $O/r06a_gpu_pytest_tail.txt
timeout 2400 bash scratch/profile_round.sh r06a > $O/profile_round.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_infer -o p -- python bench.py --mode infer --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-side-cases > $O/prof_infer.log 2>&1
cp $(find $O/prof_infer -name p_kernel_stats.csv | head -1) $O/r06a_bench_infer_b256_kernel_stats.csv
timeout 400 python bench.py --mode infer --precision fp16 --steps 20 --warmup 3 --no-cpu-baseline --no-side-cases 2>/dev/null | tail -1 > $O/r06a_bench_infer_b256_fp16.json
