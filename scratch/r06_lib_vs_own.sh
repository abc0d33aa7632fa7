#!/bin/bash
# effective clock of both arms: GRBM_GUI_ACTIVE / kernel duration per launch (one rocprofv3 --pmc pass per arm, no other trace domain)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r06_lib; mkdir -p $O
timeout 600 python scratch/r06_lib_vs_own.py time > $O/time.txt 2>&1
for arm in lib own; do
  timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/$arm -o p -- python scratch/r06_lib_vs_own.py $arm > $O/$arm.log 2>&1
done
python - "$O" <<'PY'
import csv, glob, json, sys
O = sys.argv[1]
def load(arm):
    cc = glob.glob(f"{O}/{arm}/**/p_counter_collection.csv", recursive=True)
    kt = glob.glob(f"{O}/{arm}/**/p_kernel_trace.csv", recursive=True)
    dur = {}
    for f in kt:
        for r in csv.DictReader(open(f)):
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
    rows = []
    for f in cc:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != "GRBM_GUI_ACTIVE":
                continue
            d = dur.get(r["Dispatch_Id"])
            if d is None and "End_Timestamp" in r:
                d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
            if d is None:
                continue
            rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"]), d[0]))
    rows.sort()
    return rows
for arm in ("lib", "own"):
    try:
        plan = json.load(open(f"{O}/plan_{arm}.json"))
    except Exception as e:
        print(arm, "no plan", e); continue
    rows = [r for r in load(arm) if ("Cijk" in r[1] or "gemm_nt256o" in r[1] or "gemm_tn256o" in r[1] or "gemm_nt256w" in r[1])]
    print(f"== {arm}: {len(rows)} GEMM dispatches, plan wants {sum(p['launches'] for p in plan)}")
    i = 0
    for p in plan:
        seg = rows[i:i + p["launches"]]; i += p["launches"]
        if not seg: break
        seg = seg[1:]                      # (first launch of a shape: cold)
        ns = sum(s[3] for s in seg) / len(seg); cyc = sum(s[2] for s in seg) / len(seg)
        print(f"  {p['kind']} M={p['M']:6d} {p['name']:6s} N={p['N']:5d} K={p['K']:5d}: {ns/1e3:8.1f} us  {p['flops']/ns/1e3:7.1f} TF/s  GRBM_GUI_ACTIVE/ns = {cyc/ns:6.3f}  kernel {seg[0][1][:48]}")
PY
