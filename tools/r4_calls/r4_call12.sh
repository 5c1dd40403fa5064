#!/bin/bash
# Round 4, call 12: hn recompute in the GRU backward step (forward does not store hn): parity under both settings, step A/B.
O=gpurun_out/r4l; mkdir -p $O
export PYTHONPATH=$PWD
for v in 1 0; do
GTOS_GRU_RECOMPUTE_HN=$v timeout 900 python -m pytest tests/test_hip_parity.py tests/test_full_size_c2.py -m gpu -q --tb=short -p no:cacheprovider -k "gru or trie or relation_encoder or full_bank or training_mode or c2_slice" > $O/tests_$v.log 2>&1; echo "recompute=$v"; tail -3 $O/tests_$v.log
done
for leg in h1 h0 h1b h0b; do
  v=1; case $leg in h0*) v=0;; esac
  GTOS_GRU_RECOMPUTE_HN=$v timeout 300 python bench.py --no-cpu-baseline --no-masks-leg --no-loader-leg --steps 20 --warmup 5 --prewarm-seconds 10 > $O/bench_$leg.json 2> $O/bench_$leg.err
  python - <<P
import json
try:
    d = json.loads(open("$O/bench_$leg.json").read().strip().splitlines()[-1])
    rows = [(r["kernel"][:28], r["avg_us"]) for r in d["roofline"]["kernels"] if "gru" in r["kernel"]]
    print("$leg", round(d["ms_per_step"], 2), "ms", {k: round(v, 2) for k, v in d["components"].items() if k != "note"}, rows)
except Exception as e:
    print("$leg failed", e); print(open("$O/bench_$leg.err").read()[-2500:])
P
done
GTOS_GRU_RECOMPUTE_HN=1 GTOS_BENCH_NO_DETAIL=1 timeout 300 python bench.py --relation-masks path --no-cpu-baseline --no-masks-leg --no-loader-leg --steps 10 --warmup 3 --prewarm-seconds 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('path masks, recompute hn', round(d['ms_per_step'],2))"
GTOS_GRU_RECOMPUTE_HN=0 GTOS_BENCH_NO_DETAIL=1 timeout 300 python bench.py --relation-masks path --no-cpu-baseline --no-masks-leg --no-loader-leg --steps 10 --warmup 3 --prewarm-seconds 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('path masks, saved hn', round(d['ms_per_step'],2))"
