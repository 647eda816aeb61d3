#!/bin/bash
# usage (GPU box, repo root): bash tools/pmc_pass.sh LABEL "COUNTER1 COUNTER2 ..." [more counter groups...]
# One rocprofv3 --pmc pass (kernel-trace only) of a 3-step bench per counter group; prints per-kernel averages.
label=$1; shift
repo=$(pwd); out=$repo/gpurun_out/pmc_$label; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "$@"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d $out/g$i -o g$i -- python $repo/bench.py $PMC_BENCH_ARGS --steps 3 --warmup 1 --no-cpu-baseline --no-stage-events > $out/g$i.log 2>&1 || { echo "group $i failed"; tail -5 $out/g$i.log; }
done
cd $repo
python - $out <<'PY'
import csv,glob,sys,re
from collections import defaultdict
acc=defaultdict(lambda: defaultdict(float)); cnt=defaultdict(lambda: defaultdict(int))
for f in glob.glob(sys.argv[1]+'/g*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k=re.sub(r"\(anonymous namespace\)::","",r["Kernel_Name"]); k=re.sub(r"^void ","",k); k=re.sub(r"\(.*","",k)
        acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[k][r["Counter_Name"]]+=1
with open(sys.argv[1]+'/summary.txt','w') as o:
    for k in acc:
        if not any(t in k for t in ('blend','geom','preprocess','radix_scatter','expand')): continue
        line=k[:48]+' '+' '.join(f'{c}={acc[k][c]/cnt[k][c]:.4g}' for c in sorted(acc[k]))
        print(line); o.write(line+'\n')
PY
rm -rf $out/g*/
