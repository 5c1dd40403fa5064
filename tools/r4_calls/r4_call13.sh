#!/bin/bash
# Round 4, call 13: concept / token encoders on the auxiliary stream (A/B), parity of the whole model; C5 with fresh batches under allocator settings.
O=gpurun_out/r4m; mkdir -p $O
export PYTHONPATH=$PWD
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_full_size_c2.py -m gpu -q --tb=short -p no:cacheprovider -k "generator or slice or full_batch or trainer or two_ranks or adam or overfit or end_to_end" > $O/tests.log 2>&1; tail -3 $O/tests.log
for leg in s1 s0 s1b s0b; do
  v=1; case $leg in s0*) v=0;; esac
  GTOS_CONCEPT_SIDE=$v GTOS_BENCH_NO_DETAIL=1 timeout 300 python bench.py --no-cpu-baseline --no-masks-leg --no-loader-leg --steps 20 --warmup 5 --prewarm-seconds 10 > $O/bench_$leg.json 2> $O/bench_$leg.err
  python - <<P
import json
try:
    d = json.loads(open("$O/bench_$leg.json").read().strip().splitlines()[-1])
    print("$leg", round(d["ms_per_step"], 2), "ms", {k: round(v, 2) for k, v in d["components"].items() if k != "note"})
except Exception as e:
    print("$leg failed", e); print(open("$O/bench_$leg.err").read()[-2500:])
P
done
for v in 1 0; do GTOS_CONCEPT_SIDE=$v GTOS_BENCH_NO_DETAIL=1 timeout 300 python bench.py --config C3 --no-cpu-baseline --no-masks-leg --no-loader-leg --steps 20 --warmup 5 --prewarm-seconds 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C3 side=$v', round(d['ms_per_step'],2))"; done
run() {
  PYTORCH_CUDA_ALLOC_CONF="$2" PYTORCH_ALLOC_CONF="$2" GTOS_BENCH_NO_DETAIL=1 $3 timeout 400 python bench.py --config C5 --fresh-batches --no-cpu-baseline --steps 8 --warmup 3 --prewarm-seconds 8 > $O/bench_C5_fresh_$1.json 2> $O/bench_C5_fresh_$1.err
  python - <<P
import json
try:
    d = json.loads(open("$O/bench_C5_fresh_$1.json").read().strip().splitlines()[-1])
    print("C5 fresh $1 [$2]", round(d["ms_per_step"], 1), "ms", d["config"]["device_memory"], d["config"]["loader"]["consumer_wait_ms_per_step"])
except Exception as e:
    print("C5 fresh $1 failed", e); print(open("$O/bench_C5_fresh_$1.err").read()[-800:])
P
}
run roundup "" env
run noroundup "" "env GTOS_BENCH_NO_ROUNDUP=1"
run expandable "expandable_segments:True" "env GTOS_BENCH_NO_ROUNDUP=1"
