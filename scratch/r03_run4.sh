#!/bin/bash
# what does the one-rank forced collective cost, and where: paired A/B + kernel trace of the forced run
export TMPDIR=/tmp
mkdir -p gpurun_out/r03d
scratch/ab.sh r03d_ab_fc "" "--force-collective" 3 | tee gpurun_out/r03d/ab_force_collective.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03d/prof_fc -o p -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-side-cases --force-collective > gpurun_out/r03d/prof_fc.log 2>&1
cp $(find gpurun_out/r03d/prof_fc -name p_kernel_stats.csv | head -1) gpurun_out/r03d/fc_kernel_stats.csv
head -30 gpurun_out/r03d/fc_kernel_stats.csv | cut -c1-150
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r03d/prof_fc/**/p_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rc = [r for r in rows if "ccl" in r["Kernel_Name"].lower() or "AllReduce" in r["Kernel_Name"]]
print("rccl kernels:", len(rc))
for r in rc[:12]:
    print(r["Kernel_Name"][:60], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, "us", "grid", r.get("Grid_Size"), "wg", r.get("Workgroup_Size"), "lds", r.get("LDS_Block_Size"), "vgpr", r.get("VGPR_Count"))
PY
