#!/bin/bash
# Round 5, call 27: the two-slot forward step kernel with its cell's arrays staged through LDS (line-wise global accesses; step_cell_staged)
# against the same kernel with the plain cell (GTOS_GRU_DBG=11): bit identity (tests), isolated launches, the step; + the d4-in-place test
O=gpurun_out/r5zc; mkdir -p $O
export PYTHONPATH=$PWD GTOS_BENCH_NO_DETAIL=1
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "ring_kernel_bit_identical or packed or relation_encoder or fused_step" -p no:cacheprovider 2>&1 | tail -3 | tee $O/tests.txt
timeout 300 python tools/bench_gru_step.py --dbuf --reps 8 --rows 434624 2>&1 | grep -v amdgpu.ids | grep "8 waves\|bit-id" | cut -c1-260 | tee $O/gru_fwd_staged_cell.txt
for rep in 1 2; do for dbg in 0 11; do
  GTOS_GRU_DBG=$dbg timeout 300 python bench.py --no-cpu-baseline --no-loader-leg --no-masks-leg --steps 15 --warmup 3 > $O/bench_dbg${dbg}_$rep.json 2> $O/bench_dbg${dbg}_$rep.err
  python - <<PY
import json
d = json.loads(open("$O/bench_dbg${dbg}_$rep.json").read().strip().splitlines()[-1])
print("cell %s run $rep: %.2f ms/step  RelationEncoder forward %.2f ms" % ("staged" if $dbg == 0 else "plain ", d["ms_per_step"], d["components"]["relation_encoder_fwd_ms"]))
PY
done; done 2>&1 | tee $O/summary.txt
