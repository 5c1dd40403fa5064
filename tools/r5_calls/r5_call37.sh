#!/bin/bash
# Round 5, call 37: the default bench line of the final code once more (another box of the pool, if the scheduler gives one)
O=gpurun_out/r5zl; mkdir -p $O
export PYTHONPATH=$PWD
timeout 600 python bench.py > $O/bench_c2_n1_final_again.json 2> $O/bench_c2_n1_final_again.err
python - <<PY
import json
d = json.loads(open("$O/bench_c2_n1_final_again.json").read().strip().splitlines()[-1])
print("C2", round(d["value"], 1), round(d["ms_per_step"], 2), d["components"], d["roofline"]["frac"], d["roofline"]["in_step"]["frac"], d["cpu_baseline"]["value"])
PY
