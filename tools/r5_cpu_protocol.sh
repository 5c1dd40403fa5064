#!/bin/bash
# SURVEY 8d's CPU protocol on the GPU box's host cores: 8 graphs, 2 warm-up + 5 timed steps of the pinned oracle (about five minutes).
O=gpurun_out/r5z; mkdir -p $O
export PYTHONPATH=$PWD
timeout 900 python bench.py --cpu-graphs 8 --cpu-steps 5 --cpu-warmup 2 --cpu-budget 600 --steps 10 --warmup 3 --no-masks-leg --no-loader-leg > $O/bench_c2_n1_cpu_protocol.json 2> $O/bench_c2_n1_cpu_protocol.err
python -c "
import json
d=json.loads(open('$O/bench_c2_n1_cpu_protocol.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['cpu_baseline'])"
