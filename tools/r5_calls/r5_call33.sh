#!/bin/bash
# Round 5, call 33: backward step launch with the row panel of its k loops one stage ahead (kloop_a2: two panel slots + one weight slot, three
# workgroups per CU kept) against the single-stage loops (GTOS_GRU_BWD_DBG=3): parity, isolated launches, the step, same box
O=gpurun_out/r5zi; mkdir -p $O
export PYTHONPATH=$PWD GTOS_BENCH_NO_DETAIL=1
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "gru or relation_encoder or packed or fused_step or input_gradient" -p no:cacheprovider 2>&1 | tail -3 | tee $O/tests.txt
for d in 0 3; do
  GTOS_GRU_BWD_DBG=$d timeout 300 python tools/bench_gru_step.py --only bwd,dinp --reps 8 2>&1 | grep -v amdgpu.ids | sed "s/^/bwd k loops $([ $d = 0 ] && echo 'panel one stage ahead' || echo 'single stage         '): /"
done | tee $O/gru_bwd_a2.txt
for rep in 1 2; do for d in 0 3; do
  GTOS_GRU_BWD_DBG=$d timeout 300 python bench.py --no-cpu-baseline --no-loader-leg --no-masks-leg --steps 15 --warmup 3 > $O/bench_${d}_$rep.json 2> $O/bench_${d}_$rep.err
  python - <<PY
import json
d = json.loads(open("$O/bench_${d}_$rep.json").read().strip().splitlines()[-1])
print("GTOS_GRU_BWD_DBG=$d run $rep: %.2f ms/step  GRU backward %.2f ms" % (d["ms_per_step"], d["components"]["relation_gru_bwd_ms"]))
PY
done; done 2>&1 | tee $O/summary.txt
