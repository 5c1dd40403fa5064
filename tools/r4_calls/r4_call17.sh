#!/bin/bash
# hipGraph replay of the training step: tests, then the bench leg at C1 and C2
O=gpurun_out/r4g; mkdir -p $O
export PYTHONPATH=$PWD
timeout 500 python -m pytest tests/test_zz_hip_graph_step.py -m gpu -q --tb=short -p no:cacheprovider -x > $O/graph_tests.log 2>&1; tail -25 $O/graph_tests.log | cut -c1-300
for c in C1 C2; do
  GTOS_BENCH_NO_DETAIL=1 timeout 300 python bench.py --config $c --no-cpu-baseline --no-loader-leg --no-masks-leg --steps 20 --warmup 3 --prewarm-seconds 4 > $O/bench_$c.json 2> $O/bench_$c.err
  python -c "
import json
try:
    d=json.loads(open('$O/bench_$c.json').read().strip().splitlines()[-1]); print('$c', round(d['ms_per_step'],2), 'ms eager;', d.get('hipgraph_replay'))
except Exception as e: print('$c failed', e); print(open('$O/bench_$c.err').read()[-1500:])"
done
