#!/bin/bash
# Round 5, call 16: the two-slot 64-k forward step (GTOS_GRU_FWD_DBUF=1) against the ring in the step, same box, twice each
O=gpurun_out/r5p; mkdir -p $O
export PYTHONPATH=$PWD GTOS_BENCH_NO_DETAIL=1
for rep in 1 2; do for db in 0 1; do
  GTOS_GRU_FWD_DBUF=$db timeout 300 python bench.py --no-cpu-baseline --no-loader-leg --no-masks-leg --steps 15 --warmup 3 > $O/bench_dbuf${db}_$rep.json 2> $O/bench_dbuf${db}_$rep.err
  python - <<PY
import json
d = json.loads(open("$O/bench_dbuf${db}_$rep.json").read().strip().splitlines()[-1])
print("DBUF=$db run $rep: %.2f ms/step" % d["ms_per_step"], d.get("components"))
PY
done; done 2>&1 | tee $O/summary.txt
