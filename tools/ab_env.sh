#!/bin/bash
# usage (GPU box, repo root): bash tools/ab_env.sh "STAGE [STAGE...]" REPS "ENV=..." ["ENV=..." ...]
# Like tools/ab.sh, but alternates ENVIRONMENT settings of the one product library (e.g. "X=0" "GSR_NO_SH_DIR=1") instead of
# builds: same box, same clocks, interleaved.  Prints ms_per_step, the median step and the named stage times per run, then
# the per-setting minimum of each column.  Extra bench flags: AB_ARGS.
stages=$1; reps=${2:-3}; shift 2
for i in $(seq $reps); do
  for e in "$@"; do
    env $e timeout -k 10 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline $AB_ARGS 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); st=d['roofline']['stage_ms']
print('$e', d['ms_per_step'], d['step_ms']['median'], ' '.join(str(st[k]) for k in '$stages'.split()))" || echo "$e FAILED"
  done
done | tee /tmp/ab_runs.txt
python - "$stages" <<'PY'
import sys
from collections import defaultdict
cols = ["ms_per_step", "median"] + sys.argv[1].split()
best = defaultdict(lambda: [1e9] * len(cols))
for line in open("/tmp/ab_runs.txt"):
    p = line.split()
    if len(p) != len(cols) + 1:
        continue
    best[p[0]] = [min(a, float(b)) for a, b in zip(best[p[0]], p[1:])]
print("min over runs:", " ".join(cols))
for k, v in best.items():
    print(f"  {k:24s}", " ".join(f"{x:.4f}" for x in v))
PY
