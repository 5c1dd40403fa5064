#!/bin/bash
# Round 4, call 6: the batched relation-projection weight gradient behind dX (GTOS_BATCH_DW): full-size parity + step A/B.
O=gpurun_out/r4f; mkdir -p $O
export PYTHONPATH=$PWD
timeout 600 python -m pytest tests/test_full_size_c2.py tests/test_hip_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "full_batch_generator or slice or training_mode or two_ranks" -s > $O/tests.log 2>&1; grep -a "C2 B=64\|passed\|failed\|Error\|assert" $O/tests.log | cut -c1-300 | head -20
for leg in dw1 dw0 dw1b dw0b; do
  v=1; case $leg in dw0*) v=0;; esac
  GTOS_BATCH_DW=$v GTOS_BENCH_NO_DETAIL=1 timeout 300 python bench.py --no-cpu-baseline --no-masks-leg --no-loader-leg --steps 20 --warmup 5 --prewarm-seconds 10 > $O/bench_$leg.json 2> $O/bench_$leg.err
  python - <<P
import json
try:
    d = json.loads(open("$O/bench_$leg.json").read().strip().splitlines()[-1])
    print("$leg", round(d["ms_per_step"], 2), "ms", {k: round(v, 2) for k, v in d["components"].items() if k != "note"})
except Exception as e:
    print("$leg failed", e); print(open("$O/bench_$leg.err").read()[-2500:])
P
done
