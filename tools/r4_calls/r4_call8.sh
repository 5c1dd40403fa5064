#!/bin/bash
# Round 4, call 8: C5 memory plan (per-direction release in the trie GRU backward, slab cap): memory + stability over repeated runs;
# parity of the trie GRU after the reorder; C2 unchanged; C1 with / without the mode-0 attention specialisation (two pairs).
O=gpurun_out/r4h; mkdir -p $O
export PYTHONPATH=$PWD
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_full_size_c2.py -m gpu -q --tb=short -p no:cacheprovider -k "trie or full_bank or full_batch or c5_slice" > $O/tests.log 2>&1; tail -3 $O/tests.log
for i in 1 2 3 4 5; do
  GTOS_BENCH_NO_DETAIL=1 timeout 400 python bench.py --config C5 --no-cpu-baseline --no-masks-leg --no-loader-leg --steps 6 --warmup 3 --prewarm-seconds 8 > $O/bench_C5_$i.json 2> $O/bench_C5_$i.err
  python - <<P
import json
try:
    d = json.loads(open("$O/bench_C5_$i.json").read().strip().splitlines()[-1])
    print("C5 run $i", round(d["ms_per_step"], 1), "ms", d["config"]["device_memory"], d["config"]["prewarm_steps"])
except Exception as e:
    print("C5 $i failed", e); print(open("$O/bench_C5_$i.err").read()[-1500:])
P
done
GTOS_SLAB_MAX_GB=64 GTOS_BENCH_NO_DETAIL=1 timeout 400 python bench.py --config C5 --no-cpu-baseline --no-masks-leg --no-loader-leg --steps 6 --warmup 3 --prewarm-seconds 8 2> /dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C5 with the slab', round(d['ms_per_step'],1), d['config']['device_memory'])"
GTOS_BENCH_NO_DETAIL=1 timeout 300 python bench.py --no-cpu-baseline --no-masks-leg --no-loader-leg --steps 20 --warmup 5 --prewarm-seconds 10 2> /dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C2', round(d['ms_per_step'],2), d['config']['device_memory'])"
for v in 1 0 1 0; do GTOS_ATTN_M0=$v GTOS_BENCH_NO_DETAIL=1 timeout 300 python bench.py --config C1 --no-cpu-baseline --no-masks-leg --no-loader-leg --steps 50 --warmup 10 --prewarm-seconds 5 2> /dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C1 m0=$v', round(d['ms_per_step'],2))"; done
