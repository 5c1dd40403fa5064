#!/bin/bash
# Round 5, call 14: wave count of the forward step in the step (call 13's box had four waves ahead for layer 0 in isolation): eight (default) /
# four / by input width (GTOS_GRU_FWD_NW=0: four for layer 0, eight for layer 1), twice each, one box.
O=gpurun_out/r5n; mkdir -p $O
export PYTHONPATH=$PWD GTOS_BENCH_NO_DETAIL=1
for rep in 1 2; do for nw in 8 4 0; do
  GTOS_GRU_FWD_NW=$nw timeout 300 python bench.py --no-cpu-baseline --no-loader-leg --no-masks-leg --steps 15 --warmup 3 > $O/bench_nw${nw}_$rep.json 2> $O/bench_nw${nw}_$rep.err
  python - <<PY
import json
d = json.loads(open("$O/bench_nw${nw}_$rep.json").read().strip().splitlines()[-1])
print("NW=$nw run $rep: %.2f ms/step" % d["ms_per_step"], {k: v for k, v in d.items() if "relation_enc" in k or "gru" in k})
PY
done; done 2>&1 | tee $O/summary.txt
