#!/bin/bash
# Round 5, call 24: the ring with FIVE slots of 32-k stages (four stages = 112 KB per CU in flight, 140 KB of LDS, eight waves) beside the
# three-slot ring and the two-slot 64-k kernel: isolated launches, bit identity; then the step (GTOS_GRU_FWD_DBUF=0 GTOS_GRU_FWD_RING=5).
O=gpurun_out/r5x; mkdir -p $O
export PYTHONPATH=$PWD GTOS_BENCH_NO_DETAIL=1
timeout 300 python tools/bench_gru_step.py --dbuf --reps 8 2>&1 | grep -v amdgpu.ids | cut -c1-200 | tee $O/gru_fwd_ring5.txt
for rep in 1 2; do for v in "1 1" "0 5"; do
  set -- $v
  GTOS_GRU_FWD_DBUF=$1 GTOS_GRU_FWD_RING=$2 timeout 300 python bench.py --no-cpu-baseline --no-loader-leg --no-masks-leg --steps 15 --warmup 3 > $O/bench_$1_$2_$rep.json 2> $O/bench_$1_$2_$rep.err
  python - <<PY
import json
d = json.loads(open("$O/bench_$1_$2_$rep.json").read().strip().splitlines()[-1])
print("DBUF=$1 RING=$2 run $rep: %.2f ms/step  RelationEncoder forward %.2f ms" % (d["ms_per_step"], d["components"]["relation_encoder_fwd_ms"]))
PY
done; done 2>&1 | tee $O/summary.txt
