#!/bin/bash
# paired A/B of two ENVIRONMENTS on one box (the bench command line is the same):  scratch/ab_env.sh <tag> "<env A>" "<env B>" [reps] [bench args]
tag=$1; A=$2; B=$3; reps=${4:-3}; args=${5:-}
mkdir -p gpurun_out/$tag
for i in $(seq 1 $reps); do
  env $A python bench.py --no-cpu-baseline --no-kernel-timing --no-side-cases --steps 20 $args > gpurun_out/$tag/a$i.json 2> gpurun_out/$tag/a$i.err
  env $B python bench.py --no-cpu-baseline --no-kernel-timing --no-side-cases --steps 20 $args > gpurun_out/$tag/b$i.json 2> gpurun_out/$tag/b$i.err
done
python - "$tag" "$A" "$B" "$reps" <<'PY'
import json, sys
tag, A, B, reps = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
rows = []
for i in range(1, reps + 1):
    line = lambda f: next(l for l in open(f).read().splitlines() if l.startswith('{"metric"'))
    a = json.loads(line(f"gpurun_out/{tag}/a{i}.json"))["ms_per_step"]
    b = json.loads(line(f"gpurun_out/{tag}/b{i}.json"))["ms_per_step"]
    rows.append((a, b))
print(f"A = env {A!r}\nB = env {B!r}")
for a, b in rows:
    print(f"  A {a:8.3f} ms   B {b:8.3f} ms   B - A {b - a:+7.3f} ms ({100 * (b / a - 1):+5.2f} %)")
d = sorted(b - a for a, b in rows)
print(f"  median paired delta {d[len(d) // 2]:+.3f} ms/step")
PY
