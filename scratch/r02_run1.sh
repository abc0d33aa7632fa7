#!/bin/bash
# round-2 first GPU pass: full GPU test suite, the three bench modes, per-launch breakdown, PMC calibration
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r02a; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 -s > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -30 $O/pytest.log
python bench.py --steps 10 --warmup 3 2>$O/bench_train.err | tail -1 > $O/bench_train.json; cat $O/bench_train.json
python bench.py --mode infer --steps 10 --warmup 3 2>$O/bench_infer.err | tail -1 > $O/bench_infer.json; cat $O/bench_infer.json
python bench.py --mode ts --steps 5 --warmup 2 2>$O/bench_ts.err | tail -1 > $O/bench_ts.json; cat $O/bench_ts.json
python bench.py --hip-graph --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing 2>$O/bench_graph.err | tail -1 > $O/bench_graph.json; cat $O/bench_graph.json
python scratch/step_breakdown.py > $O/step_breakdown.txt 2>&1; tail -45 $O/step_breakdown.txt
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/calib_$c -o p -- python scratch/fetch_calib.py > /dev/null 2>&1
  python - "$O/calib_$c" <<'PY'
import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + "/**/p_counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f[0])):
    k = (r["Kernel_Name"][:60], r["Counter_Name"]); agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
for k, (n, v) in sorted(agg.items()):
    print(f"{k[0]:62s} {k[1]:11s} n={n} avg={v/n/1024:.1f} MiB (counter is KB)")
PY
done
