#!/bin/bash
# closing kernel trace of the default bench command on the round's final code (kernel stats, by grid, one step as an ordered kernel list)
O=gpurun_out/r4pz; mkdir -p $O
export PYTHONPATH=$PWD
R=$PWD
cd /tmp && export TMPDIR=/tmp && GTOS_BENCH_NO_DETAIL=1 timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o trace -- python $R/bench.py --relation-masks node --steps 4 --warmup 1 --no-cpu-baseline --no-masks-leg --no-loader-leg --prewarm-seconds 5 > $R/$O/bench_line_under_rocprof.json 2> $R/$O/bench_rocprof.err
cd $R
DB=$(find $O/prof -name "*.db" | head -1)
python tools/rocpd_stats.py $DB $O/kernel_stats.csv > /dev/null
python tools/rocpd_stats.py $DB $O/kernel_stats_by_grid.csv --by-grid > /dev/null
python tools/rocpd_sequence.py $DB --step 3 > $O/step_sequence.txt; head -1 $O/step_sequence.txt
python tools/rocpd_timeline.py $DB 8 > $O/timeline.txt 2>&1; head -1 $O/timeline.txt
find $O/prof -name "*stats*" | head; for f in $(find $O/prof -name "*kernel_stats*.csv" | head -1); do cp $f $O/rocprofv3_kernel_stats.csv; done
rm -rf $O/prof
head -8 $O/kernel_stats.csv | cut -c1-140; grep "rel_attn_fwd_kernel" $O/kernel_stats_by_grid.csv | cut -c1-160
python -c "
import json; d=json.loads(open('$O/bench_line_under_rocprof.json').read().strip().splitlines()[-1]); print('bench under rocprof', round(d['ms_per_step'],2), d['roofline']['frac'], d['roofline']['avg_us'])"
