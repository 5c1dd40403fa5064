#!/bin/bash
O=gpurun_out/r3j; mkdir -p $O
export PYTHONPATH=$PWD
timeout 300 python -m pytest tests/test_hip_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "fresh or prefetcher or bf16_vs_golden" 2>&1 | tail -3
for i in 1 2 3 4 5; do
  GTOS_BENCH_NO_DETAIL=1 timeout 200 python bench.py --no-cpu-baseline --fresh-batches --workers 4 --steps 20 > $O/fresh_$i.json 2> $O/fresh_$i.err; rc=$?
  python -c "
import json
try:
    d=json.load(open('$O/fresh_$i.json')); print('fresh run $i rc=$rc', round(d['ms_per_step'],2), d['config']['loader']['consumer_wait_ms_per_step'])
except Exception as e: print('fresh run $i rc=$rc failed', e)"
  tail -3 $O/fresh_$i.err | cut -c1-300
done
