#!/bin/bash
O=gpurun_out/r4m; mkdir -p $O
export PYTHONPATH=$PWD
timeout 200 python -m pytest tests/test_hip_parity.py -m gpu -q --tb=short -p no:cacheprovider -s -k "projection_recomputed" 2>&1 | grep -v amdgpu.ids | tail -6 | cut -c1-250
run() { n=$1; c=$2; shift; shift; env "$@" GTOS_BENCH_NO_DETAIL=1 timeout 400 python bench.py --config $c --no-cpu-baseline --no-loader-leg --no-masks-leg > $O/bench_$n.json 2> $O/bench_$n.err
  python -c "
import json
try:
    d=json.loads(open('$O/bench_$n.json').read().strip().splitlines()[-1]); print('$n', round(d['ms_per_step'],2), 'ms', round(d['value'],1), d['config']['device_memory'])
except Exception as e: print('$n failed', e); print(open('$O/bench_$n.err').read()[-800:])"; }
run c5_proj_recompute C5 GTOS_PROJ_RECOMPUTE=1
