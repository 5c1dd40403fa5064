#!/bin/bash
# the default bench line after the switch of its default to the reference's dropout semantics, and the 2-rank bench test
O=gpurun_out/r4zzz; mkdir -p $O
export PYTHONPATH=$PWD
( time timeout 600 python bench.py > $O/bench_c2_n1.json 2> $O/bench_c2_n1.err ) 2> $O/bench_c2_n1.time; python - <<P
import json
try:
    d=json.loads(open("$O/bench_c2_n1.json").read().strip().splitlines()[-1]); print("bench", round(d["value"],1), round(d["ms_per_step"],2), d["components"]); print(d["roofline"]["frac"], d["roofline"]["in_step"]["frac"], d["cpu_baseline"]["value"], d["node_masks"], d["loader_in_loop"]["ms_per_step"], d["config"]["device_memory"]); print(d["config"]["relation_gru"])
except Exception as e: print("bench failed", e); print(open("$O/bench_c2_n1.err").read()[-2000:])
P
grep real $O/bench_c2_n1.time
timeout 400 python -m pytest tests/test_hip_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "bench" 2>&1 | tail -3
