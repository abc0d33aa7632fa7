#!/bin/bash
# round-4 evidence: scratch/profile_round.sh r04b (bench lines, rocprofv3 kernel stats, PMC traffic + calibration, SQ/GRBM utilisation) +
# rocprofv3 kernel stats of the inference (configs[1]) and 30 s training (configs[3] per-GPU shape) runs, kernels serialized
export TMPDIR=/tmp
bash scratch/profile_round.sh r04b > gpurun_out/r04b_round.log 2>&1
O=gpurun_out/r04b
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_infer -o p -- python bench.py --mode infer --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-side-cases --serial-kernels > $O/prof_infer.log 2>&1
cp $(find $O/prof_infer -name p_kernel_stats.csv | head -1) $O/r04b_bench_infer_b256_kernel_stats.csv
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_t30 -o p -- python bench.py --frames 1876 --batch 128 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-side-cases --serial-kernels > $O/prof_t30.log 2>&1
cp $(find $O/prof_t30 -name p_kernel_stats.csv | head -1) $O/r04b_bench_train30s_b128_serial_kernel_stats.csv
rm -rf $O/prof_infer $O/prof_t30 $O/prof $O/pmc_* $O/calib_*
tail -25 gpurun_out/r04b_round.log
head -8 $O/r04b_bench_infer_b256_kernel_stats.csv | cut -c1-150
head -12 $O/r04b_bench_train30s_b128_serial_kernel_stats.csv | cut -c1-150
