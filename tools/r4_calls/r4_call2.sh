#!/bin/bash
# Round 4, call 2: the fp32 residual stream -- bf16 parity numbers, the whole GPU suite, a bench line with and without it.
O=gpurun_out/r4b; mkdir -p $O
export PYTHONPATH=$PWD
timeout 300 python -m pytest tests/test_hip_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "bf16" -s > $O/bf16_parity.log 2>&1
grep -a "bf16 vs\|passed\|failed\|Error\|error" $O/bf16_parity.log | head -40
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > $O/gpu_tests.log 2>&1
tail -15 $O/gpu_tests.log
GTOS_BENCH_NO_DETAIL=1 timeout 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_fp32stream.json 2> $O/bench_fp32stream.err
GTOS_FP32_STREAM=0 GTOS_BENCH_NO_DETAIL=1 timeout 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --prewarm-seconds 5 > $O/bench_bf16stream.json 2> $O/bench_bf16stream.err
python - <<P
import json
for n in ("fp32stream", "bf16stream"):
    try:
        d = json.loads(open("$O/bench_%s.json" % n).read().strip().splitlines()[-1])
        print(n, round(d["ms_per_step"], 2), "ms", d["config"]["loss_first"], d["config"]["loss_last"])
    except Exception as e:
        print(n, "failed", e); print(open("$O/bench_%s.err" % n).read()[-1500:])
P
