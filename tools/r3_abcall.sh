#!/bin/bash
O=gpurun_out/r3o; mkdir -p $O
export PYTHONPATH=$PWD
( time timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $O/gpu_tests.log 2>&1 ) 2> $O/gpu_tests.time; tail -4 $O/gpu_tests.log; grep real $O/gpu_tests.time
GTOS_BENCH_VERBOSE=1 GTOS_BENCH_NO_DETAIL=1 timeout 300 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print('bench', round(d['value'],1), round(d['ms_per_step'],3), d['config']['prewarm_steps'])"; grep prewarm $O/bench.err | head -12
cd /tmp && export TMPDIR=/tmp && GTOS_BENCH_NO_DETAIL=1 timeout 300 rocprofv3 --kernel-trace -d $OLDPWD/$O/prof -o trace -- python $OLDPWD/bench.py --steps 4 --warmup 1 --prewarm-seconds 0 --no-cpu-baseline > $OLDPWD/$O/bench_rocprof.json 2> $OLDPWD/$O/bench_rocprof.err
cd $OLDPWD
DB=$(find $O/prof -name "*.db" | head -1)
python tools/rocpd_sequence.py $DB --step 3 > $O/step_sequence.txt; head -1 $O/step_sequence.txt; grep -c "direct_copy" $O/step_sequence.txt; grep "transpose_batch" $O/step_sequence.txt | head -2
rm -rf $O/prof
