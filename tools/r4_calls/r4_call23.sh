#!/bin/bash
# the step with the reference's dropout semantics (bench.py's default now): does the auxiliary stream pay there?
O=gpurun_out/r4s; mkdir -p $O
export PYTHONPATH=$PWD
run() { n=$1; shift; env "$@" GTOS_BENCH_NO_DETAIL=1 timeout 200 python bench.py --no-cpu-baseline --no-loader-leg --no-masks-leg --steps 15 --warmup 3 --prewarm-seconds 4 > $O/bench_$n.json 2> $O/bench_$n.err
  python -c "
import json
d=json.loads(open('$O/bench_$n.json').read().strip().splitlines()[-1]); print('$n', round(d['ms_per_step'],2), 'ms', round(d['value'],1), d['config']['device_memory'])"; }
run default GTOS_X=0
run gru_side_off GTOS_GRU_SIDE=0
run all_side_off GTOS_SIDE_STREAMS=0
