#!/bin/bash
# Round 5, call 29: C3 (launches of at most 100 k rows) with the eight-wave and the four-wave form of the two-slot forward step kernel, three runs each
O=gpurun_out/r5ze; mkdir -p $O
export PYTHONPATH=$PWD GTOS_BENCH_NO_DETAIL=1
for rep in 1 2 3; do for db in 1 4; do
  GTOS_GRU_FWD_DBUF=$db timeout 300 python bench.py --config C3 --no-cpu-baseline --no-loader-leg --no-masks-leg --steps 20 --warmup 5 --prewarm-seconds 5 > $O/bench_C3_db${db}_$rep.json 2> $O/bench_C3_db${db}_$rep.err
  python - <<PY
import json
d = json.loads(open("$O/bench_C3_db${db}_$rep.json").read().strip().splitlines()[-1])
print("C3 DBUF=$db run $rep: %.2f ms/step  RelationEncoder forward %.2f ms" % (d["ms_per_step"], d["components"]["relation_encoder_fwd_ms"]))
PY
done; done 2>&1 | tee $O/summary.txt
