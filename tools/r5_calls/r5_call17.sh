#!/bin/bash
# Round 5, call 17: the two-slot 64-k forward step with four waves (128-row panels, 80 KB: two workgroups per CU) beside the eight-wave
# form and the ring: isolated launches, then the step (GTOS_GRU_FWD_DBUF = 0 ring / 1 eight waves / 4 four waves), same box.
O=gpurun_out/r5q; mkdir -p $O
export PYTHONPATH=$PWD GTOS_BENCH_NO_DETAIL=1
timeout 300 python tools/bench_gru_step.py --dbuf --reps 8 2>&1 | grep -v amdgpu.ids | tee $O/gru_fwd_dbuf.txt
for rep in 1 2; do for db in 0 1 4; do
  GTOS_GRU_FWD_DBUF=$db timeout 300 python bench.py --no-cpu-baseline --no-loader-leg --no-masks-leg --steps 15 --warmup 3 > $O/bench_dbuf${db}_$rep.json 2> $O/bench_dbuf${db}_$rep.err
  python - <<PY
import json
d = json.loads(open("$O/bench_dbuf${db}_$rep.json").read().strip().splitlines()[-1])
print("DBUF=$db run $rep: %.2f ms/step  RelationEncoder forward %.2f ms" % (d["ms_per_step"], d["components"]["relation_encoder_fwd_ms"]))
PY
done; done 2>&1 | tee $O/summary.txt
