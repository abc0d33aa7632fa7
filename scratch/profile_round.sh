#!/bin/bash
# full evidence run on the GPU box: bench JSONs (train / infer / ts) + rocprofv3 kernel stats (serialized) + PMC traffic
# of the dominant GEMM with a FETCH/WRITE calibration on known byte counts + SQ utilisation counters of the hot kernels
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
TAG=${1:-r03a}; O=gpurun_out/$TAG; mkdir -p $O
timeout 600 python bench.py --steps 10 --warmup 3 2>/dev/null | tail -1 > $O/${TAG}_bench_train_b256.json
timeout 400 python bench.py --mode infer --steps 20 --warmup 3 2>/dev/null | tail -1 > $O/${TAG}_bench_infer_b256.json
timeout 400 python bench.py --mode ts --steps 5 --warmup 2 2>/dev/null | tail -1 > $O/${TAG}_bench_ts_b128.json
timeout 400 python bench.py --frames 1876 --batch 128 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/${TAG}_bench_train30s_b128.json
timeout 400 python bench.py --mode infer --frames 1876 --batch 64 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/${TAG}_bench_infer30s_b64.json
timeout 400 python bench.py --mode infer --precision bf16x3 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/${TAG}_bench_infer_b256_bf16x3.json
timeout 600 python bench.py --mode infer --precision fp32 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/${TAG}_bench_infer_b256_fp32.json
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing --serial-kernels > $O/prof.log 2>&1
cp $(find $O/prof -name p_kernel_stats.csv | head -1) $O/${TAG}_bench_train_b256_serial_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --serial-kernels > /dev/null 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/calib_$c -o p -- python scratch/fetch_calib.py > /dev/null 2>&1
done
python - "$O" "$TAG" <<'PY'
import csv, json, collections, glob, sys
O, TAG = sys.argv[1], sys.argv[2]
def agg(pattern):
    out = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for f in glob.glob(pattern, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            a = out[k][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
    return out
bench = agg(O + "/pmc_*/**/p_counter_collection.csv")
calib = agg(O + "/calib_*/**/p_counter_collection.csv")
# calibration: counter KB vs known bytes
known = {"layernorm_fwd_kernel": (74240 * 768 * 4, 74240 * 768 * 2),
         "gemm_nt256w_kernel<unsigned short, 0>": None}
cal = {}
for k, cs in calib.items():
    cal[k[:70]] = {c: {"launches": n, "avg_bytes": v / n * 1024} for c, (n, v) in cs.items()}
ln = next((v for k, v in cal.items() if "layernorm_fwd" in k), None)
fetch_factor = None
if ln and "FETCH_SIZE" in ln:
    fetch_factor = (74240 * 768 * 4) / ln["FETCH_SIZE"]["avg_bytes"]
write_factor = (74240 * 768 * 2) / ln["WRITE_SIZE"]["avg_bytes"] if ln and "WRITE_SIZE" in ln else None
nt = {k: v for k, v in bench.items() if "gemm_nt256w" in k or "gemm_nt256o" in k}
steps = 4   # 1 warm-up + 3 steps profiled
# per GEMM CALL (bench.py times a call, i.e. the 256-row-tile launch plus, where the last partial round is split off, the
# 128-row-tile launch behind it): 91 calls per training step
calls = 91 * steps
kernel_launches = sum(cs["FETCH_SIZE"][0] for cs in nt.values()) if nt else 0
launches = calls
fetch = sum(cs["FETCH_SIZE"][1] for cs in nt.values()) * 1024 / calls
write = sum(cs["WRITE_SIZE"][1] for cs in nt.values()) * 1024 / calls
out = {"source": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python bench.py --steps 3 --warmup 1 "
                 f"--no-cpu-baseline --no-kernel-timing --serial-kernels  (scratch/profile_round.sh {TAG})",
       "kernel": "gemm_nt256o_kernel (bf16, one wave per SIMD) + gemm_nt256w_kernel<bf16, ...> (128-row tail tiles, row-dot form): all launches of a step",
       "launches_per_step": launches // steps, "kernel_launches_per_step": kernel_launches // steps,
       "fetch_bytes_per_launch_raw": fetch, "write_bytes_per_launch": write,
       "fetch_correction": "x2 on gfx950 for wide coalesced reads (MI355X_MICROARCH.md, HBM section)",
       "calibration": {"what": "scratch/fetch_calib.py under the same two --pmc passes: counter bytes vs known bytes",
                       "fetch_known_over_counter_layernorm_fwd": fetch_factor, "write_known_over_counter_layernorm_fwd": write_factor,
                       "kernels": cal},
       "traffic_bytes_per_launch": 2 * fetch + write,
       "per_step_GB_raw": {k[:70]: {c: round(v * 1024 / 1e9 / steps, 2) for c, (n, v) in cs.items()} for k, cs in
                           sorted(bench.items(), key=lambda kv: -sum(x[1] for x in kv[1].values()))[:14]}}
json.dump(out, open(f"{O}/{TAG}_pmc_traffic.json", "w"), indent=1)
print(json.dumps({k: out[k] for k in ("launches_per_step", "fetch_bytes_per_launch_raw", "write_bytes_per_launch", "traffic_bytes_per_launch")}))
print("calibration factors (known / counter): fetch", fetch_factor, "write", write_factor)
for k, v in cal.items(): print("  calib", k[:60], {c: round(x["avg_bytes"] / 1e6, 1) for c, x in v.items()}, "MB")
PY
bash scratch/pmc_util.sh $TAG > $O/pmc_util.txt 2>&1
cp $O/mfma_util.json $O/${TAG}_mfma_util.json
head -22 $O/${TAG}_bench_train_b256_serial_kernel_stats.csv | cut -c1-160
for f in $O/${TAG}_bench_*.json; do echo $f; python -c "import json,sys; d=json.load(open('$f')); print(d['value'], d['ms_per_step'], d.get('roofline',{}).get('frac'), d.get('attention_set',{}).get('mfma_frac'), d.get('kernel_ms_per_step'))"; done
