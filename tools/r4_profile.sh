#!/bin/bash
# Round 4 profile run: kernel trace of the bench (kernel stats, one step as an ordered kernel list), whole-step PMC traffic, PMC traffic of
# the relation-attention kernels per operand mode.  Summaries under gpurun_out/r4p/ (copied to profiles/ by hand).
O=gpurun_out/r4p; mkdir -p $O
export PYTHONPATH=$PWD
R=$PWD
cd /tmp && export TMPDIR=/tmp && GTOS_BENCH_NO_DETAIL=1 timeout 300 rocprofv3 --kernel-trace -d $R/$O/prof -o trace -- python $R/bench.py --relation-masks node --steps 4 --warmup 1 --no-cpu-baseline --no-masks-leg --no-loader-leg --prewarm-seconds 5 > $R/$O/bench_line_under_rocprof.json 2> $R/$O/bench_rocprof.err
cd $R
DB=$(find $O/prof -name "*.db" | head -1)
python tools/rocpd_stats.py $DB $O/kernel_stats.csv > /dev/null
python tools/rocpd_stats.py $DB $O/kernel_stats_by_grid.csv --by-grid > /dev/null
python tools/rocpd_sequence.py $DB --step 3 > $O/step_sequence.txt; head -1 $O/step_sequence.txt
python tools/rocpd_timeline.py $DB 8 > $O/timeline.txt 2>&1; head -1 $O/timeline.txt
rm -rf $O/prof
ROUND=r4 timeout 700 bash tools/pmc_step.sh > $O/pmc_step.log 2>&1; tail -3 $O/pmc_step.log
ROUND=r4 timeout 700 bash tools/pmc_rel_attn.sh > $O/pmc_rel_attn.log 2>&1; tail -12 $O/pmc_rel_attn.log
