#!/bin/bash
O=gpurun_out/r4m; mkdir -p $O
export PYTHONPATH=$PWD
run() { n=$1; c=$2; shift; shift; env "$@" GTOS_BENCH_NO_DETAIL=1 timeout 400 python bench.py --config $c --no-cpu-baseline --no-loader-leg --no-masks-leg > $O/bench_$n.json 2> $O/bench_$n.err
  python -c "
import json
try:
    d=json.loads(open('$O/bench_$n.json').read().strip().splitlines()[-1]); print('$n', round(d['ms_per_step'],2), 'ms', round(d['value'],1), d['config']['device_memory'])
except Exception as e: print('$n failed', e); print(open('$O/bench_$n.err').read()[-800:])"; }
run c5_auto_expandable C5 PYTORCH_CUDA_ALLOC_CONF=expandable_segments:True GTOS_BENCH_NO_ROUNDUP=1
