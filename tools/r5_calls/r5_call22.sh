#!/bin/bash
# Round 5, call 22: double-buffered k loops in the BACKWARD step launch (kloop_dbuf, both roles; two workgroups per CU) against the library
# of the previous commit (single-stage loops, three workgroups per CU; forward step: the ring) on one box: parity, isolated launches, the step.
O=gpurun_out/r5v; mkdir -p $O
export PYTHONPATH=$PWD GTOS_BENCH_NO_DETAIL=1
L=gtos_amd/csrc
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "gru or relation_encoder or packed or fused_step or input_gradient" -p no:cacheprovider 2>&1 | tail -3 | tee $O/tests.txt
cp $L/libgtos_hip.so /tmp/libnew.so
for which in new head new head; do
  if [ $which = head ]; then cp $L/libgtos_hip_head.so $L/libgtos_hip.so; else cp /tmp/libnew.so $L/libgtos_hip.so; fi
  timeout 300 python tools/bench_gru_step.py --only bwd,dinp --reps 8 2>&1 | grep -v amdgpu.ids | sed "s/^/$which: /"
  timeout 300 python bench.py --no-cpu-baseline --no-loader-leg --no-masks-leg --steps 15 --warmup 3 > $O/bench_$which.json 2> $O/bench_$which.err
  python - <<PY
import json
d = json.loads(open("$O/bench_$which.json").read().strip().splitlines()[-1])
print("$which: %.2f ms/step" % d["ms_per_step"], d["components"])
PY
done 2>&1 | tee $O/summary.txt
cp /tmp/libnew.so $L/libgtos_hip.so
