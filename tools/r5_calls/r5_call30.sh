#!/bin/bash
# Round 5, call 30: the forward step with two slots of activation rows and THREE of weight rows (80 KB per CU in flight instead of 56;
# gru_step_fwd_a2w3_kernel): isolated launches, bit identity; then the step (GTOS_GRU_FWD_A2W3=1) against the default, same box.
O=gpurun_out/r5zf; mkdir -p $O
export PYTHONPATH=$PWD GTOS_BENCH_NO_DETAIL=1
timeout 300 python tools/bench_gru_step.py --dbuf --reps 8 2>&1 | grep -v amdgpu.ids | grep "8 waves\|three slots\|bit-id" | cut -c1-150 | tee $O/gru_fwd_a2w3.txt
for rep in 1 2; do for v in 0 1; do
  GTOS_GRU_FWD_A2W3=$v timeout 300 python bench.py --no-cpu-baseline --no-loader-leg --no-masks-leg --steps 15 --warmup 3 > $O/bench_a2w3_${v}_$rep.json 2> $O/bench_a2w3_${v}_$rep.err
  python - <<PY
import json
d = json.loads(open("$O/bench_a2w3_${v}_$rep.json").read().strip().splitlines()[-1])
print("A2W3=$v run $rep: %.2f ms/step  RelationEncoder forward %.2f ms" % (d["ms_per_step"], d["components"]["relation_encoder_fwd_ms"]))
PY
done; done 2>&1 | tee $O/summary.txt
