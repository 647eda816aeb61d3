#!/bin/bash
# usage: tools_stage.sh LABEL [ENV=VAL ...]  -- run a short bench and print the stage table
label=$1; shift
env "$@" timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_$label.json 2> gpurun_out/bench_$label.err
python - "$label" <<'PY'
import sys,json
l=sys.argv[1]
try:
    d=json.loads(open(f"gpurun_out/bench_{l}.json").read().strip().splitlines()[-1])
    print(l, d["value"], d["ms_per_step"], d["roofline"]["stage_ms"])
except Exception as e:
    print(l, "FAILED", e); print(open(f"gpurun_out/bench_{l}.err").read()[-2000:])
PY
