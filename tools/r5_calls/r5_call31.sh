#!/bin/bash
# Round 5, call 31: A two / W three slots against A three / W two (88 KB in flight) against the two-slot kernel: isolated, then the step (three legs, twice)
O=gpurun_out/r5zg; mkdir -p $O
export PYTHONPATH=$PWD GTOS_BENCH_NO_DETAIL=1
timeout 300 python tools/bench_gru_step.py --dbuf --reps 8 --rows 434624 2>&1 | grep -v amdgpu.ids | grep "rows  434624" | grep "8 waves\|slots +\|bit-id" | cut -c1-150 | tee $O/gru_fwd_a2w3_a3w2.txt
for rep in 1 2; do for v in 0 1 2; do
  GTOS_GRU_FWD_A2W3=$v timeout 300 python bench.py --no-cpu-baseline --no-loader-leg --no-masks-leg --steps 15 --warmup 3 > $O/bench_${v}_$rep.json 2> $O/bench_${v}_$rep.err
  python - <<PY
import json
d = json.loads(open("$O/bench_${v}_$rep.json").read().strip().splitlines()[-1])
print("GTOS_GRU_FWD_A2W3=$v run $rep: %.2f ms/step  RelationEncoder forward %.2f ms" % (d["ms_per_step"], d["components"]["relation_encoder_fwd_ms"]))
PY
done; done 2>&1 | tee $O/summary.txt
