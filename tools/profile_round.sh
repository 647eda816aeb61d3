#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/profile_round.sh LABEL
# Writes under gpurun_out/prof_LABEL/: the default bench line, the rocprofv3 kernel-trace stats of a 10-step bench, the
# FETCH_SIZE / WRITE_SIZE counter passes (separate runs, kernel-trace only) summarised by tools/summarize_pmc.py, the SQ
# counter passes (instruction counts, wave / wait / active cycles, GRBM clock) summarised by tools/summarize_sq.py, and the
# vector-instruction issue-cost calibration (tools/valu_rate).  Copy what is to be judged from there into profiles/.
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
SQ1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAVES"
SQ2="SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH"
SQ3="GRBM_GUI_ACTIVE SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS SQ_INSTS_SMEM"
i=0
for grp in "$SQ1" "$SQ2" "$SQ3"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d $out/sq$i -o sq$i -- python $repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-stage-events > $out/sq$i.log 2>&1
done
cd $repo
[ -x tools/valu_rate ] && timeout -k 10 200 tools/valu_rate > $out/${label}_valu_issue_costs.txt 2>&1
stats=$(find $out/kt -name '*kernel_stats.csv' | head -1)
cp "$stats" $out/${label}_kernel_stats_bench_steps10.csv
mkdir -p $out/summary
cp profiles/pmc_traffic.json $out/summary/ 2>/dev/null || true
python tools/summarize_pmc.py $label C3 $(find $out/fetch -name '*counter_collection.csv' | head -1) $(find $out/write -name '*counter_collection.csv' | head -1) $out/summary
cp profiles/sq_counters.json $out/summary/ 2>/dev/null || true
python tools/summarize_sq.py $label C3 $out/summary $(find $out/sq1 $out/sq2 $out/sq3 -name '*counter_collection.csv')
rm -rf $out/kt $out/fetch $out/write $out/sq1 $out/sq2 $out/sq3      # keep the merge small: the summaries are what gets committed
# the counters now on file belong to this build: run the default bench once more so its line carries roofline.traffic and
# roofline_valu (bench.py quotes them only for the build they were measured on)
cp $out/summary/pmc_traffic.json $out/summary/sq_counters.json profiles/
timeout -k 10 400 python bench.py > $out/${label}_bench_line.json 2>> $out/bench.err
tail -c 900 $out/${label}_bench_line.json; echo
head -8 $out/${label}_kernel_stats_bench_steps10.csv | cut -c1-160
