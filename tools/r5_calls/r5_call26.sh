#!/bin/bash
# Round 5, call 26: backward of the packed path writes d4 IN PLACE over the saved gates (GTOS_GRU_D4_INPLACE, default 1): parity, C2 step and
# memory with / without, C5 memory.
O=gpurun_out/r5zb; mkdir -p $O
export PYTHONPATH=$PWD GTOS_BENCH_NO_DETAIL=1
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_full_size_c2.py -m gpu -q -x -k "packed or relation_encoder or training_mode or fused_step or input_gradient or c2_slice" -p no:cacheprovider 2>&1 | tail -3 | tee $O/tests.txt
for rep in 1 2; do for ip in 1 0; do
  GTOS_GRU_D4_INPLACE=$ip timeout 300 python bench.py --no-cpu-baseline --no-loader-leg --no-masks-leg --steps 15 --warmup 3 > $O/bench_ip${ip}_$rep.json 2> $O/bench_ip${ip}_$rep.err
  python - <<PY
import json
d = json.loads(open("$O/bench_ip${ip}_$rep.json").read().strip().splitlines()[-1])
print("in place=$ip run $rep: %.2f ms/step  GRU backward %.2f ms" % (d["ms_per_step"], d["components"]["relation_gru_bwd_ms"]), d["config"]["device_memory"])
PY
done; done 2>&1 | tee $O/summary.txt
timeout 400 python bench.py --config C5 --steps 8 --warmup 3 --no-cpu-baseline --no-masks-leg --no-loader-leg --prewarm-seconds 8 > $O/bench_C5.json 2> $O/bench_C5.err
python -c "
import json
d=json.loads(open('$O/bench_C5.json').read().strip().splitlines()[-1]); print('C5', round(d['ms_per_step'],2), d['config']['device_memory'])" | tee -a $O/summary.txt
