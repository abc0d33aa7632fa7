#!/bin/bash
# round-5 evidence on the current build: whole GPU suite, scratch/profile_round.sh <tag> (bench lines, rocprofv3 kernel stats, PMC traffic +
# calibration, SQ/GRBM utilisation), rocprofv3 kernel stats of the inference and 30 s training runs, the default driver line
export TMPDIR=/tmp
TAG=${1:-r05a}
O=gpurun_out/$TAG
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/${TAG}_gpu_pytest.log 2>&1; echo "pytest exit $?" >> $O/${TAG}_gpu_pytest.log
tail -4 $O/${TAG}_gpu_pytest.log > $O/${TAG}_gpu_pytest_tail.txt; cat $O/${TAG}_gpu_pytest_tail.txt
bash scratch/profile_round.sh $TAG > gpurun_out/${TAG}_round.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_infer -o p -- python bench.py --mode infer --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-side-cases --serial-kernels > $O/prof_infer.log 2>&1
cp $(find $O/prof_infer -name p_kernel_stats.csv | head -1) $O/${TAG}_bench_infer_b256_kernel_stats.csv
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_t30 -o p -- python bench.py --frames 1876 --batch 128 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-side-cases --serial-kernels > $O/prof_t30.log 2>&1
cp $(find $O/prof_t30 -name p_kernel_stats.csv | head -1) $O/${TAG}_bench_train30s_b128_serial_kernel_stats.csv
rm -rf $O/prof_infer $O/prof_t30 $O/prof $O/pmc_* $O/calib_*
S=$(date +%s); timeout 900 python bench.py > $O/${TAG}_bench_default_line.json 2> $O/bench_default.err; echo "default line wall $(( $(date +%s) - S )) s" | tee $O/${TAG}_bench_default_wall.txt
tail -25 gpurun_out/${TAG}_round.log
head -6 $O/${TAG}_bench_infer_b256_kernel_stats.csv | cut -c1-150
head -8 $O/${TAG}_bench_train30s_b128_serial_kernel_stats.csv | cut -c1-150
tail -c 600 $O/${TAG}_bench_default_line.json
