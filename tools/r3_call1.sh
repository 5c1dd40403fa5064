#!/bin/bash
# Round-3 GPU call 1: full gpu test suite (incl. the new full-size C2 tests), round-start bench, loader-in-the-loop bench,
# dropout-semantics A/B.  Everything writes under gpurun_out/r3a/.
O=gpurun_out/r3a; mkdir -p $O
export PYTHONPATH=$PWD
( time timeout 1500 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider -s > $O/gpu_tests.log 2>&1 ) 2> $O/gpu_tests.time
tail -5 $O/gpu_tests.log
timeout 300 python bench.py --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err; tail -c 600 $O/bench_c2.json
for w in 2 4 8; do
  GTOS_BENCH_NO_DETAIL=1 timeout 300 python bench.py --no-cpu-baseline --fresh-batches --workers $w --steps 20 > $O/bench_c2_fresh_w$w.json 2> $O/bench_c2_fresh_w$w.err
  python - <<P
import json
try:
    d=json.load(open("$O/bench_c2_fresh_w$w.json")); print("fresh w=$w", d["value"], d["ms_per_step"], d["config"]["loader"])
except Exception as e: print("fresh w=$w failed", e)
P
done
timeout 900 python tools/dropout_ab.py --steps 2000 --out $O/dropout_ab.json > $O/dropout_ab.log 2>&1; tail -25 $O/dropout_ab.log
