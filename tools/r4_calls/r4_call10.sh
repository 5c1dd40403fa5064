#!/bin/bash
O=gpurun_out/r4j; mkdir -p $O
export PYTHONPATH=$PWD
timeout 300 python -m pytest tests/test_hip_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "batched_relation_projection" > $O/tests.log 2>&1; tail -3 $O/tests.log
for i in 1 2 3; do
GTOS_BENCH_NO_DETAIL=1 timeout 400 python bench.py --config C5 --no-cpu-baseline --no-masks-leg --no-loader-leg --steps 8 --warmup 3 --prewarm-seconds 8 2> $O/c5_$i.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C5 empty_cache after first window', round(d['ms_per_step'],1), d['config']['device_memory'])" || tail -5 $O/c5_$i.err
done
GTOS_BENCH_NO_DETAIL=1 timeout 300 python bench.py --no-cpu-baseline --no-masks-leg --no-loader-leg --steps 20 --warmup 5 --prewarm-seconds 10 2> /dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C2', round(d['ms_per_step'],2), d['config']['device_memory'])"
