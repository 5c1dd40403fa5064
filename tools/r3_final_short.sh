#!/bin/bash
# Round-3 closing run on the last GPU minutes: full gpu suite, default bench (with cpu_baseline), kernel trace of the bench, C5.
O=gpurun_out/r3y; mkdir -p $O
export PYTHONPATH=$PWD
( time timeout 240 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -s > $O/gpu_tests.log 2>&1 ) 2> $O/gpu_tests.time
tail -3 $O/gpu_tests.log; grep real $O/gpu_tests.time
timeout 150 python bench.py > $O/bench_c2_n1.json 2> $O/bench_c2_n1.err; python - <<P
import json
try:
    d=json.load(open("$O/bench_c2_n1.json")); print("bench", d["value"], d["ms_per_step"], d["components"]); print(d["roofline"]["frac"], d["roofline"]["avg_us"], d["cpu_baseline"]["value"])
except Exception as e: print("bench failed", e); print(open("$O/bench_c2_n1.err").read()[-2000:])
P
R=$PWD
cd /tmp && export TMPDIR=/tmp && timeout 120 rocprofv3 --kernel-trace -d $R/$O/prof -o trace -- python $R/bench.py --steps 4 --warmup 1 --prewarm-seconds 4 --no-cpu-baseline > $R/$O/bench_line_under_rocprof.json 2> $R/$O/bench_rocprof.err
cd $R
DB=$(find $O/prof -name "*.db" | head -1)
python tools/rocpd_stats.py $DB $O/kernel_stats.csv > /dev/null
python tools/rocpd_stats.py $DB $O/kernel_stats_by_grid.csv --by-grid > /dev/null
python tools/rocpd_sequence.py $DB --step 3 > $O/step_sequence.txt; head -1 $O/step_sequence.txt
python tools/rocpd_timeline.py $DB 8 > $O/timeline.txt 2>&1
rm -rf $O/prof
GTOS_BENCH_NO_DETAIL=1 timeout 100 python bench.py --config C5 --steps 6 --warmup 2 --prewarm-seconds 6 --no-cpu-baseline > $O/bench_C5.json 2> $O/bench_C5.err
python -c "
import json
d=json.load(open('$O/bench_C5.json')); print('C5', round(d['value'],1), round(d['ms_per_step'],2), d['config']['device_memory'], d['roofline']['frac'])"
