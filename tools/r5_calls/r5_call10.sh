#!/bin/bash
# Round 5, tenth GPU call: the last changes -- loader leg at N > 1 (two ranks on one GPU), the two-stream test, the bench line reading the round-5 counters.
O=gpurun_out/r5j; mkdir -p $O
export PYTHONPATH=$PWD
timeout 1200 python -m pytest tests/test_hip_parity.py -m gpu -q -p no:cacheprovider -k "bench_ or two_stream or slice_full_depth or c3_slice" > $O/tests.log 2>&1
echo "tests rc=$? $(tail -1 $O/tests.log)"; grep -E "^FAILED|^ERROR" $O/tests.log | head
timeout 400 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/r5j/bench_default.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['traffic_source'][:40], d['roofline']['in_step']['frac'], d['roofline']['in_step']['traffic'])
P
