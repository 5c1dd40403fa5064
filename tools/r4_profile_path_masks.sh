#!/bin/bash
# kernel trace of the training step with the reference's dropout semantics (masks per (path, position): the per-row RelationEncoder)
O=gpurun_out/r4q; mkdir -p $O
export PYTHONPATH=$PWD
R=$PWD
cd /tmp && export TMPDIR=/tmp && GTOS_BENCH_NO_DETAIL=1 timeout 300 rocprofv3 --kernel-trace -d $R/$O/prof -o trace -- python $R/bench.py --relation-masks path --steps 4 --warmup 1 --no-cpu-baseline --no-masks-leg --no-loader-leg --prewarm-seconds 3 > $R/$O/bench_line_under_rocprof.json 2> $R/$O/bench_rocprof.err
cd $R
DB=$(find $O/prof -name "*.db" | head -1)
python tools/rocpd_stats.py $DB $O/kernel_stats.csv > /dev/null
python tools/rocpd_stats.py $DB $O/kernel_stats_by_grid.csv --by-grid > /dev/null
python tools/rocpd_sequence.py $DB --step 3 > $O/step_sequence.txt; head -1 $O/step_sequence.txt
rm -rf $O/prof
head -22 $O/kernel_stats.csv | cut -c1-150
