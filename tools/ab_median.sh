#!/bin/bash
# usage (GPU box, repo root): bash tools/ab_median.sh REPS "ENV=..." ["ENV=..." ...]
# Alternates environment settings of the one product library; no stage events (kernels back to back, as in production); prints
# ms_per_step and the step_ms median / p10 per run and the minimum and median over runs per setting.  Extra bench flags: AB_ARGS.
reps=${1:-5}; shift 1
for i in $(seq $reps); do
  for e in "$@"; do
    env $e timeout -k 10 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-stage-events $AB_ARGS 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$e', d['ms_per_step'], d['step_ms']['median'], d['step_ms']['p10'])" || echo "$e FAILED"
  done
done | tee /tmp/abm_runs.txt
python - <<'PY'
from collections import defaultdict
import statistics
runs = defaultdict(list)
for line in open("/tmp/abm_runs.txt"):
    p = line.split()
    if len(p) == 4:
        runs[p[0]].append([float(x) for x in p[1:]])
print("per setting: min / median over runs of (ms_per_step, step median, step p10)")
for k, v in runs.items():
    cols = list(zip(*v))
    print(f"  {k:24s} min " + " ".join(f"{min(c):.4f}" for c in cols) + "   median " + " ".join(f"{statistics.median(c):.4f}" for c in cols))
PY
