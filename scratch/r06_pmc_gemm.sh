#!/bin/bash
# HBM-side reads of the wide GEMMs with and without column panels: rocprofv3 --pmc FETCH_SIZE (own pass), per launch; x2 on gfx950
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r06_pmc; mkdir -p $O
for shape in "3072 768 pair" "3072 768 mul" "3072 768 none" "2304 768 none" "768 3072 none" "768 768 none"; do
  for pan in 0 -1; do
    set -- $shape
    d=$O/f_$1_$2_$3_p$pan
    MAEST_GEMM_PANEL=$pan timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $d -o p -- python scratch/r06_fc1_one.py $1 $2 $3 > /dev/null 2>&1
    python - "$d" "$shape" "$pan" <<'PY'
import csv, glob, sys
d, shape, pan = sys.argv[1], sys.argv[2], sys.argv[3]
N, K, form = shape.split(); N = int(N); K = int(K); M = 74240
tot, n = 0.0, 0
for f in glob.glob(d + "/**/p_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_nt256o" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            tot += float(r["Counter_Value"]); n += 1
alg = (M * K + N * K) * 2 + (M * N * 2 if form == "mul" else 0)
if n:
    print(f"N={N} K={K} {form:5s} panel {pan:>2s}: 2 x FETCH_SIZE = {2*tot*1024/n/1e6:8.1f} MB per launch ({n} launches); operands {alg/1e6:7.1f} MB -> {2*tot*1024/n/alg:5.2f} x")
else:
    print(f"N={N} K={K} {form} panel {pan}: no launches found")
PY
  done
done
