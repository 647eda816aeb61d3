#!/bin/bash
# usage (GPU box, repo root): [GSR_LIB=...] bash tools/kt.sh [bench args]  -- rocprofv3 kernel-trace of a 10-step bench, per-kernel averages
repo=$(pwd); out=$repo/gpurun_out/kt_tmp; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o kt -- python $repo/bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > $out/log.txt 2>&1 || tail -5 $out/log.txt
cd $repo
f=$(find $out -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, re, sys
tot = 0.0
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    n = re.sub(r"\(anonymous namespace\)::", "", r["Name"]); n = re.sub(r"^void ", "", n); n = re.sub(r"\(.*", "", n)
    rows.append((n, int(r["Calls"]), float(r["AverageNs"]) / 1000.0, float(r["TotalDurationNs"]) / 1000.0))
steps = max(1, next((c for n, c, a, t in rows if n.startswith("preprocess_kernel")), 13))
for n, c, a, t in rows:
    if n.startswith("at::") or "rocclr_copy" in n: continue
    print(f"{n[:58]:58s} {c / steps:5.1f}/step {a:8.1f} us  {t / steps:8.1f} us/step")
    tot += t / steps
print(f"{'sum':58s} {tot:30.1f} us/step")
PY
cp "$f" $repo/gpurun_out/kt_last_kernel_stats.csv
rm -rf $out
