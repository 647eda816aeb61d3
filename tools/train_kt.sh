#!/bin/bash
# usage (GPU box, repo root): bash tools/train_kt.sh [ITERATIONS] [extra train.py args]  -- rocprofv3 kernel trace of the Lego trainer
# (examples/train.py on data/lego, the reference schedule): GPU microseconds per iteration by kernel, and their sum against the
# trainer's own wall clock per iteration (the difference is host time the GPU waits for).
iters=${1:-1500}; shift
repo=$(pwd); out=$repo/gpurun_out/tkt_tmp; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o kt -- python $repo/examples/train.py --dataset $repo/data/lego --views 8 \
    --iterations $iters --gaussians 5000 --print-interval 500 "$@" > $out/log.txt 2>&1 || tail -5 $out/log.txt
cd $repo
grep "^trained" $out/log.txt
f=$(find $out -name "*kernel_stats.csv" | head -1)
python3 - "$f" "$iters" <<'PY'
import csv, re, sys
iters = int(sys.argv[2])
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    n = re.sub(r"\(anonymous namespace\)::", "", r["Name"]); n = re.sub(r"^void ", "", n); n = re.sub(r"\(.*", "", n)
    rows.append((n, int(r["Calls"]), float(r["AverageNs"]) / 1000.0, float(r["TotalDurationNs"]) / 1000.0))
tot = 0.0
for n, c, a, t in sorted(rows, key=lambda r: -r[3]):
    tot += t / iters
    if t / iters >= 1.0:
        print(f"{n[:70]:70s} {c / iters:6.2f}/iter {a:8.1f} us  {t / iters:8.1f} us/iter")
print(f"{'sum of all kernels':70s} {tot:34.1f} us/iter")
PY
cp "$f" $repo/gpurun_out/train_kt_kernel_stats.csv
rm -rf $out
