#!/bin/bash
# Vector-memory path counters of the full-line NT kernel on a long-K problem (scratch/probe/ablw_FULL, random operands):
# is the wave stalled handing LDS-DMA requests to the texture-address unit (FIFO full), is the TA stalled by the cache,
# how busy is it.  Separate --pmc passes, kernel-trace only.  -> gpurun_out/$TAG/pmc_vmem.txt
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
TAG=${1:-r02}; O=gpurun_out/$TAG; mkdir -p $O
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_INSTS_FLAT SQ_INSTS_MFMA"
P2="SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL"
P3="TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum"
P4="TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum"
P5="TA_FLAT_READ_LDS_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum"
P6="TCP_PENDING_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum"
P7="TCC_HIT_sum TCC_MISS_sum"
P8="GRBM_GUI_ACTIVE GRBM_TA_BUSY"
i=0
for p in "$P1" "$P2" "$P3" "$P4" "$P5" "$P6" "$P7" "$P8"; do
  i=$((i+1))
  (cd scratch/probe && timeout 300 rocprofv3 --kernel-trace --pmc $p --output-format csv -d $R/$O/pmc_vmem_$i -o p -- ./ablw_FULL FULL) > $O/pmc_vmem_$i.log 2>&1 || tail -5 $O/pmc_vmem_$i.log
done
python - "$O" <<'PY'
import csv, sys, glob, collections
O = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob(O + "/pmc_vmem_*/**/p_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_nt256w" not in r["Kernel_Name"]: continue
        key = (r["Grid_Size"] if "Grid_Size" in r else "?")
        a = agg[key][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
with open(O + "/pmc_vmem.txt", "w") as out:
    for key, cs in sorted(agg.items()):
        d = {c: v / n for c, (n, v) in cs.items()}
        line = f"grid {key}: " + "  ".join(f"{c}={v:.4g}" for c, v in sorted(d.items()))
        print(line); out.write(line + "\n")
        wc = d.get("SQ_WAVE_CYCLES")
        if wc:
            fr = {c: d[c] / wc for c in d if c.startswith("SQ_") and c not in ("SQ_WAVE_CYCLES",) and "INSTS" not in c}
            line = "   / SQ_WAVE_CYCLES: " + "  ".join(f"{c}={v:.3f}" for c, v in sorted(fr.items()))
            print(line); out.write(line + "\n")
PY
