#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/profile_round.sh LABEL
# Writes under gpurun_out/prof_LABEL/: the default bench line, the rocprofv3 kernel-trace stats of a 10-step bench, and the
# FETCH_SIZE / WRITE_SIZE counter passes (separate runs, kernel-trace only) summarised by tools/summarize_pmc.py.
set -e
label=$1
repo=$(pwd)
out=$repo/gpurun_out/prof_$label
mkdir -p $out
timeout -k 10 400 python bench.py > $out/${label}_bench_line.json 2> $out/bench.err
tail -c 600 $out/${label}_bench_line.json; echo
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- python $repo/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/kt.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o fetch -- python $repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-stage-events > $out/fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -o write -- python $repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-stage-events > $out/write.log 2>&1
cd $repo
stats=$(find $out/kt -name '*kernel_stats.csv' | head -1)
cp "$stats" $out/${label}_kernel_stats_bench_steps10.csv
mkdir -p $out/summary
cp profiles/pmc_traffic.json $out/summary/ 2>/dev/null || true
python tools/summarize_pmc.py $label C3 $(find $out/fetch -name '*counter_collection.csv' | head -1) $(find $out/write -name '*counter_collection.csv' | head -1) $out/summary
rm -rf $out/kt $out/fetch $out/write      # keep the merge small: the summaries are what gets committed
head -8 $out/${label}_kernel_stats_bench_steps10.csv | cut -c1-160
