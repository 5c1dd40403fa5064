#!/bin/bash
# Round 4, call 5: mask-injection tests of the RelationEncoder's training mode (reference masks / trie-shared masks), loader default test,
# the whole GPU suite, the default bench line with its two secondary legs (other mask mode, loader in the loop).
O=gpurun_out/r4e; mkdir -p $O
export PYTHONPATH=$PWD
timeout 300 python -m pytest tests/test_hip_parity.py tests/test_zzz_hip_relbatch.py -m gpu -q --tb=short -p no:cacheprovider -k "training_mode or loader_default or trie_gru_dropout" -s > $O/new_tests.log 2>&1; grep -a "MEASURED\|passed\|failed\|Error\|assert" $O/new_tests.log | head -20
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $O/gpu_tests.log 2>&1; tail -8 $O/gpu_tests.log
( time timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time; tail -3 $O/bench_default.time
python - <<P
import json
try:
    d = json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
    print("default", round(d["ms_per_step"], 2), "ms", d["components"], d["roofline"]["frac"], d["roofline"]["in_step"]["frac"])
    print("   relation_gru:", d["config"]["relation_gru"])
    print("   reference_masks", d.get("reference_masks"))
    print("   loader_in_loop", d.get("loader_in_loop"))
    print("   cpu", d.get("cpu_baseline", {}).get("value"))
except Exception as e:
    print("bench failed", e); print(open("$O/bench_default.err").read()[-2500:])
P
