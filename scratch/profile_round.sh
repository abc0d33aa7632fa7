#!/bin/bash
# full evidence run on the GPU box: bench JSONs + rocprofv3 kernel stats (serialized) + PMC traffic for the GEMM
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
TAG=${1:-r01d}
python bench.py --steps 10 --warmup 3 2>&1 | tail -1 > gpurun_out/${TAG}_bench_train_b256.json
python bench.py --mode infer --steps 10 --warmup 3 2>&1 | tail -1 > gpurun_out/${TAG}_bench_infer_b256.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_prof -o p -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing --serial-kernels > gpurun_out/${TAG}_prof.log 2>&1
cp gpurun_out/${TAG}_prof/p_kernel_stats.csv gpurun_out/${TAG}_bench_train_b256_serial_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/${TAG}_pmc_$c -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --serial-kernels > /dev/null 2>&1
done
python - <<PY
import csv, json, collections
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f"gpurun_out/${TAG}_pmc_{c}/p_counter_collection.csv")):
        k = r["Kernel_Name"].split("(")[0][:60]
        agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
    for k, (n, v) in agg.items():
        out.setdefault(k, {})[c] = {"launches": n, "avg_KB": v / n, "total_GB": v * 1024 / 1e9}
json.dump(out, open("gpurun_out/${TAG}_pmc_summary.json", "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -sum(x["total_GB"] for x in kv[1].values()))[:12]:
    print(k, {c: (x["launches"], round(x["avg_KB"] / 1024, 1)) for c, x in v.items()})
PY
head -25 gpurun_out/${TAG}_bench_train_b256_serial_kernel_stats.csv | cut -c1-150
cat gpurun_out/${TAG}_bench_train_b256.json; cat gpurun_out/${TAG}_bench_infer_b256.json
