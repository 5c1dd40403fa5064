#!/bin/bash
# where the host time of a launch-bound step goes: cProfile over C1 steps
O=gpurun_out/r4h; mkdir -p $O
export PYTHONPATH=$PWD
GTOS_BENCH_NO_DETAIL=1 timeout 200 python -m cProfile -o $O/c1.prof bench.py --config C1 --no-cpu-baseline --no-loader-leg --no-masks-leg --steps 200 --warmup 3 --prewarm-seconds 2 > $O/bench_C1.json 2> $O/bench_C1.err
python - <<P
import pstats, json
d=json.loads(open("$O/bench_C1.json").read().strip().splitlines()[-1]); print("C1 under cProfile", round(d["ms_per_step"],2), "ms/step")
p=pstats.Stats("$O/c1.prof"); p.sort_stats("tottime").print_stats(45)
P
