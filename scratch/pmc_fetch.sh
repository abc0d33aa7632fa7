#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | tr ' ' '_')
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmcf_$n -o p -- python ${1:-scratch/tn_one.py} > /dev/null 2>&1
  f=$R/gpurun_out/pmcf_$n/p_counter_collection.csv
  echo "== $c"
  python - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    k = (r["Kernel_Name"][:70], r["Counter_Name"])
    agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
for k, (n, v) in sorted(agg.items()):
    if "maest" in k[0]: print(f"{k[0]:72s} {k[1]:14s} n={n} avg={v/n:.5g}")
PY
done
