#!/bin/bash
# round-over-round on ONE box: the round-2 tree (scratch/r02_tree = git archive of the round-2 head, built here) against the
# working tree, the same bench command alternately
reps=${1:-3}
mkdir -p gpurun_out/r03_rounds
for i in $(seq 1 $reps); do
  (cd scratch/r02_tree && python bench.py --no-cpu-baseline --no-kernel-timing --steps 20 2>/dev/null | grep '^{"metric"' > ../../gpurun_out/r03_rounds/r02_$i.json)
  python bench.py --no-cpu-baseline --no-kernel-timing --no-side-cases --steps 20 2>/dev/null | grep '^{"metric"' > gpurun_out/r03_rounds/r03_$i.json
  (cd scratch/r02_tree && python bench.py --mode infer --no-cpu-baseline --no-kernel-timing --steps 20 2>/dev/null | grep '^{"metric"' > ../../gpurun_out/r03_rounds/r02i_$i.json)
  python bench.py --mode infer --no-cpu-baseline --no-kernel-timing --no-side-cases --steps 20 2>/dev/null | grep '^{"metric"' > gpurun_out/r03_rounds/r03i_$i.json
done
python - $reps <<'PY'
import json, sys
reps = int(sys.argv[1])
for tag, name in (("", "training step configs[2]"), ("i", "inference configs[1]")):
    print(name)
    for i in range(1, reps + 1):
        a = json.load(open(f"gpurun_out/r03_rounds/r02{tag}_{i}.json")); b = json.load(open(f"gpurun_out/r03_rounds/r03{tag}_{i}.json"))
        print(f"  round 2 {a['ms_per_step']:8.3f} ms ({a['value']:8.1f} clips/s)   round 3 {b['ms_per_step']:8.3f} ms ({b['value']:8.1f} clips/s)   {100 * (b['ms_per_step'] / a['ms_per_step'] - 1):+5.2f} %")
PY
