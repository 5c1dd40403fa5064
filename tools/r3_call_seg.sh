#!/bin/bash
O=gpurun_out/r3s; mkdir -p $O
export PYTHONPATH=$PWD
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "segment_sum or trie or relation_encoder or full_depth" > $O/tests.log 2>&1; tail -5 $O/tests.log
run() { # name, env...
  n=$1; shift
  env "$@" GTOS_BENCH_NO_DETAIL=1 timeout 300 python bench.py --no-cpu-baseline --steps 20 > $O/$n.json 2> $O/$n.err
  python - <<P
import json
try:
    d=json.load(open("$O/$n.json")); print("$n", round(d["value"],1), round(d["ms_per_step"],3), d["components"]["relation_gru_bwd_ms"], d["config"]["prewarm_steps"], d["config"]["device_memory"])
except Exception as e: print("$n failed", e); print(open("$O/$n.err").read()[-1500:])
P
}
run warm GTOS_SEG_STREAM=1
run stream_a GTOS_SEG_STREAM=1
run chunk_a GTOS_SEG_STREAM=0
run stream_b GTOS_SEG_STREAM=1
run chunk_b GTOS_SEG_STREAM=0
timeout 300 python bench.py --no-cpu-baseline --steps 6 > $O/detail.json 2> $O/detail.err
python - <<P
import json
d=json.load(open("$O/detail.json"))
for k in d["roofline"]["kernels"]:
    if "seg_sum" in k["kernel"]: print(k)
P
GTOS_BENCH_NO_DETAIL=1 timeout 200 python bench.py --no-cpu-baseline --fresh-batches --workers 2 --steps 30 > $O/bench_c2_fresh_p2.json 2> $O/bench_c2_fresh_p2.err
python - <<P
import json
try:
    d=json.load(open("$O/bench_c2_fresh_p2.json")); print("fresh procs=2", d["value"], d["ms_per_step"], d["config"]["loader"], d["config"]["device_memory"])
except Exception as e: print("fresh procs=2 failed", e); print(open("$O/bench_c2_fresh_p2.err").read()[-1500:])
P
GTOS_BENCH_NO_DETAIL=1 timeout 300 python bench.py --config C5 --steps 6 --warmup 3 --no-cpu-baseline > $O/bench_C5.json 2> $O/bench_C5.err
python -c "
import json
d=json.load(open('$O/bench_C5.json')); print('C5', round(d['value'],1), round(d['ms_per_step'],2), d['components'], d['config']['device_memory'], d['config']['prewarm_steps'])"
