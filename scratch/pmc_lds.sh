#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
for c in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU"; do
  n=$(echo $c | tr ' ' '_')
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_$n -o p -- python scratch/tn_one.py > /dev/null 2>&1
  f=$(ls $R/gpurun_out/pmc_$n/p_counter_collection.csv 2>/dev/null | head -1)
  echo "== $c -> $f"
  python - "$f" <<'PY'
import csv, sys, collections
f = sys.argv[1]
if not f: sys.exit()
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    k = (r["Kernel_Name"][:60], r["Counter_Name"])
    agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
for k, (n, v) in sorted(agg.items()):
    if "gemm" in k[0]: print(f"{k[0]:62s} {k[1]:32s} n={n} avg={v/n:.4g}")
PY
done
