#!/bin/bash
# usage (GPU box, repo root): bash tools/ab.sh STAGE [REPS]   -- alternate ab/lib_base.so and ab/lib_new.so on THIS box, print the stage time
stage=$1; reps=${2:-3}
for i in $(seq $reps); do
  for l in base new; do
    GSR_LIB=$PWD/ab/lib_$l.so timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$l', d['ms_per_step'], d['roofline']['stage_ms']['$stage'])"
  done
done
