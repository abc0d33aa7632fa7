#!/bin/bash
# SQ counters of the hot kernels (separate --pmc passes, kernel-trace only) -> gpurun_out/$TAG/mfma_util.json
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
TAG=${1:-r02}; O=gpurun_out/$TAG; mkdir -p $O
rocprofv3 -L > $O/counters.txt 2>&1
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAVES"
P3="GRBM_GUI_ACTIVE GRBM_COUNT"
i=0
for p in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $p --output-format csv -d $O/pmc_util_$i -o p -- python scratch/kern_mix.py 2 > $O/pmc_util_$i.log 2>&1 || tail -5 $O/pmc_util_$i.log
done
python - "$O" <<'PY'
import csv, sys, glob, json, collections
O = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob(O + "/pmc_util_*/**/p_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("maest::", "")[:70]
        a = agg[k][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
out = {"_units": "per-launch averages over `launches` launches of scratch/kern_mix.py (B = 256, N = 290, bf16, random data). "
                 "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs): the fraction of SIMD cycles, at "
                 "the clock the kernel actually ran at, in which the matrix pipe was busy (SQ_VALU_MFMA_BUSY_CYCLES = 32 x "
                 "SQ_INSTS_MFMA for 32x32x16 bf16, 16 x for the 16x16x32 form gemm_nt256o_kernel uses since round 6; GRBM_GUI_ACTIVE is summed over the 8 XCDs).  *_frac_of_wave_cycles: SQ "
                 "wave-state counters over SQ_WAVE_CYCLES (quad-cycles).  lds_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE."}
for k, cs in agg.items():
    if not any(s in k for s in ("gemm", "attn", "layernorm")): continue
    d = {c: v / n for c, (n, v) in cs.items()}
    d["launches"] = max(n for n, v in cs.values())
    if d.get("GRBM_GUI_ACTIVE"):
        d["mfma_util"] = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * d["GRBM_GUI_ACTIVE"] / 8.0)
    if d.get("SQ_WAVE_CYCLES"):
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS"):
            if c in d: d[c + "_frac_of_wave_cycles"] = d[c] / d["SQ_WAVE_CYCLES"]
    if d.get("SQ_LDS_IDX_ACTIVE"):
        d["lds_conflict_frac"] = d.get("SQ_LDS_BANK_CONFLICT", 0) / d["SQ_LDS_IDX_ACTIVE"]
    out[k] = d
json.dump(out, open(O + "/mfma_util.json", "w"), indent=1)
for k, d in out.items():
    if k[0] != "_": print(k[:60], {c: round(v, 4) for c, v in d.items() if "frac" in c or c == "mfma_util"}, d["launches"])
PY
