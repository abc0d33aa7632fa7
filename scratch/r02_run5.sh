#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r02j; mkdir -p $O
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "attention or gemm" > $O/pytest_sel.log 2>&1; tail -3 $O/pytest_sel.log
timeout 300 python scratch/attn_bench.py 2>&1 | tee $O/attn_bench.txt
timeout 400 python scratch/store_policy_sweep.py 2>&1 | tee $O/store_policy.txt
for pol in 0 1; do
  MAEST_GEMM_STORE=$pol timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch_pol$pol -o p -- python scratch/fetch_calib.py > /dev/null 2>&1
  python - "$O/fetch_pol$pol" $pol <<'PY'
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/p_counter_collection.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "gemm_nt256w" in r["Kernel_Name"]: print("policy", sys.argv[2], "grid", r.get("Grid_Size"), "FETCH_SIZE x2 =", round(float(r["Counter_Value"]) * 2 * 1024 / 1e6, 1), "MB")
PY
done
for pol in 0 1; do MAEST_GEMM_STORE=$pol timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('store policy $pol:', d['value'], d['ms_per_step'], d['roofline']['frac'], d['kernel_ms_per_step'])"; done
