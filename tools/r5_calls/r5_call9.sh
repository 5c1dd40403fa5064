#!/bin/bash
# Round 5, ninth GPU call: the other BASELINE configs in the DEFAULT mode (C5's memory first), smoke.
O=gpurun_out/r5i; mkdir -p $O
export PYTHONPATH=$PWD
timeout 300 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log | cut -c1-400
for c in C5 C3 C1; do
  GTOS_BENCH_NO_DETAIL=1 timeout 500 python bench.py --config $c --steps 8 --warmup 3 --no-cpu-baseline --prewarm-seconds 8 > $O/bench_$c.json 2> $O/bench_$c.err
  python -c "
import json
try:
    d=json.loads(open('$O/bench_$c.json').read().strip().splitlines()[-1]); print('$c', round(d['value'],1), round(d['ms_per_step'],2), d['components'], d['config']['device_memory'], (d.get('node_masks') or d.get('reference_masks') or {}).get('ms_per_step'), (d.get('loader_in_loop') or {}).get('ms_per_step'))
except Exception as e: print('$c failed', e); print(open('$O/bench_$c.err').read()[-1500:])"
done
