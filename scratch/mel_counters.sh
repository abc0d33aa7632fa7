#!/bin/bash
# SQ wave-state counters and HBM bytes (FETCH_SIZE / WRITE_SIZE, separate --pmc passes, kernel-trace only) of logmel_kernel at batch 256 x 10 s
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/mel_counters; mkdir -p $O
cat > /tmp/mel_run.py <<'PY'
import sys, torch
sys.path.insert(0, ".")
import bench
r = bench.time_mel_kernel("cuda", 256, 160000, reps=5)
print(r)
PY
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES"
P3="GRBM_GUI_ACTIVE GRBM_COUNT"
P4="FETCH_SIZE"
P5="WRITE_SIZE"
i=0
for p in "$P1" "$P2" "$P3" "$P4" "$P5"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $p --output-format csv -d $O/p$i -o p -- python /tmp/mel_run.py > $O/p$i.log 2>&1 || tail -5 $O/p$i.log
done
python - "$O" <<'PY'
import csv, sys, glob, json, collections
O = sys.argv[1]
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(O + "/p*/**/p_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "logmel_kernel" not in r["Kernel_Name"]: continue
        a = agg[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
d = {c: v / n for c, (n, v) in agg.items()}
d["launches"] = max(n for n, v in agg.values())
if d.get("SQ_WAVE_CYCLES"):
    for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_WAIT_INST_LDS"):
        if c in d: d[c + "_frac_of_wave_cycles"] = d[c] / d["SQ_WAVE_CYCLES"]
if d.get("SQ_LDS_IDX_ACTIVE"): d["lds_conflict_frac"] = d.get("SQ_LDS_BANK_CONFLICT", 0) / d["SQ_LDS_IDX_ACTIVE"]
if d.get("SQ_WAVES"): d["valu_insts_per_wave"] = d.get("SQ_INSTS_VALU", 0) / d["SQ_WAVES"]; d["lds_insts_per_wave"] = d.get("SQ_INSTS_LDS", 0) / d["SQ_WAVES"]
alg = 256 * (160000 * 4 + 96 * 626 * 4)
d["algorithmic_bytes"] = alg
if "FETCH_SIZE" in d: d["fetch_bytes_x2_gfx950"] = 2 * d["FETCH_SIZE"] * 1024
if "WRITE_SIZE" in d: d["write_bytes"] = d["WRITE_SIZE"] * 1024
if "FETCH_SIZE" in d and "WRITE_SIZE" in d: d["hbm_bytes_over_algorithmic"] = (d["fetch_bytes_x2_gfx950"] + d["write_bytes"]) / alg
d["_units"] = ("per-launch averages of logmel_kernel (batch 256 x 160000 samples); FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them, fetch x2 on gfx950 "
               "(MI355X_MICROARCH.md, calibrated on LayerNorm's known bytes in profiles/r05f_pmc_traffic.json); *_frac_of_wave_cycles over SQ_WAVE_CYCLES")
json.dump(d, open(O + "/logmel_counters.json", "w"), indent=1)
print(json.dumps(d, indent=1))
PY
