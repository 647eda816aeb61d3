#!/bin/bash
# usage (GPU box, repo root): bash tools/ab.sh "STAGE [STAGE...]" REPS NAME [NAME...]
# Alternates the builds ab/lib_NAME.so on THIS box (same device, same clocks, interleaved) and prints, per run, ms_per_step,
# the median step and the named stage times; then the per-build minimum of each column.  Extra bench flags: AB_ARGS.
stages=$1; reps=${2:-3}; shift 2
for i in $(seq $reps); do
  for l in "$@"; do
    GSR_LIB=$PWD/ab/lib_$l.so timeout -k 10 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline $AB_ARGS 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); st=d['roofline']['stage_ms']
print('$l', d['ms_per_step'], d['step_ms']['median'], ' '.join(str(st[k]) for k in '$stages'.split()))" || echo "$l FAILED"
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
    print(f"  {k:12s}", " ".join(f"{x:.4f}" for x in v))
PY
