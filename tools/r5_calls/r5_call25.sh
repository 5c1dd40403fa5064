#!/bin/bash
# Round 5, call 25: the default bench line and C1 once more on another box of the pool (box-to-box spread of the headline and of the launch-bound config)
O=gpurun_out/r5y; mkdir -p $O
export PYTHONPATH=$PWD
timeout 600 python bench.py > $O/bench_c2_n1_box2.json 2> $O/bench_c2_n1_box2.err
GTOS_BENCH_NO_DETAIL=1 timeout 300 python bench.py --config C1 --steps 8 --warmup 3 --no-cpu-baseline --prewarm-seconds 8 --graph-leg > $O/bench_C1_box2.json 2> $O/bench_C1_box2.err
python - <<PY
import json
d = json.loads(open("$O/bench_c2_n1_box2.json").read().strip().splitlines()[-1])
print("C2", round(d["value"], 1), round(d["ms_per_step"], 2), d["components"], d["roofline"]["frac"], d["roofline"]["in_step"]["frac"], d["cpu_baseline"]["value"], d["cpu_baseline"].get("cpu"))
c = json.loads(open("$O/bench_C1_box2.json").read().strip().splitlines()[-1])
print("C1", round(c["ms_per_step"], 2), (c.get("hipgraph_replay") or {}).get("ms_per_step"))
PY
lscpu | grep -i "model name\|^CPU(s)\|MHz" | head -4
