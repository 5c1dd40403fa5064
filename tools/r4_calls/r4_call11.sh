#!/bin/bash
# Round 4, call 11: double-buffered k loop of the GRU backward step kernel: parity tests, step A/B, kernel durations from the detail pass.
O=gpurun_out/r4k; mkdir -p $O
export PYTHONPATH=$PWD
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_full_size_c2.py -m gpu -q --tb=short -p no:cacheprovider -k "gru or trie or relation_encoder or full_bank or training_mode" > $O/tests.log 2>&1; tail -3 $O/tests.log
for leg in p1 p0 p1b p0b; do
  v=1; case $leg in p0*) v=0;; esac
  GTOS_GRU_BWD_PIPE=$v timeout 300 python bench.py --no-cpu-baseline --no-masks-leg --no-loader-leg --steps 20 --warmup 5 --prewarm-seconds 10 > $O/bench_$leg.json 2> $O/bench_$leg.err
  python - <<P
import json
try:
    d = json.loads(open("$O/bench_$leg.json").read().strip().splitlines()[-1])
    rows = [r for r in d["roofline"]["kernels"] if "gru_step_bwd" in r["kernel"]]
    print("$leg", round(d["ms_per_step"], 2), "ms", {k: round(v, 2) for k, v in d["components"].items() if k != "note"}, [(r["avg_us"], r["frac"]) for r in rows])
except Exception as e:
    print("$leg failed", e); print(open("$O/bench_$leg.err").read()[-2500:])
P
done
