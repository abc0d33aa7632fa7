#!/bin/bash
# bench line + rocprofv3 kernel stats of the SAME serialized configuration on ONE box (the agreement check of DESIGN 5)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
TAG=${1:-r02c}; O=gpurun_out/$TAG; mkdir -p $O
timeout 400 python bench.py --steps 10 --warmup 3 2>/dev/null | tail -1 > $O/${TAG}_bench_train_b256.json
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof2 -o p -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing --serial-kernels > $O/prof2.log 2>&1
cp $(find $O/prof2 -name p_kernel_stats.csv | head -1) $O/${TAG}_bench_train_b256_serial_kernel_stats.csv
timeout 300 python bench.py --mode infer --steps 10 --warmup 3 2>/dev/null | tail -1 > $O/${TAG}_bench_infer_b256.json
head -4 $O/${TAG}_bench_train_b256_serial_kernel_stats.csv | cut -c1-150
python -c "
import json
for f in ('train','infer'):
    d=json.load(open('$O/${TAG}_bench_%s_b256.json' % f)); r=d['roofline']
    print(f, d['value'], d['ms_per_step'], r['frac'], r['achieved'], r['launches_per_step'], r['avg_launch_ms'], r['ms_per_step'], d.get('attention_set',{}).get('mfma_frac'))
"
