#!/bin/bash
# Round 5, call 18: what the BACKWARD step launch is made of (GTOS_GRU_BWD_DBG: 1 = no k loops, 2 = the k loops alone), 434,624 rows
O=gpurun_out/r5r; mkdir -p $O
export PYTHONPATH=$PWD
for d in 0 1 2; do
  GTOS_GRU_BWD_DBG=$d timeout 300 python tools/bench_gru_step.py --only bwd,dinp --reps 8 2>&1 | grep -v amdgpu.ids | sed "s/^/bwd_dbg=$d (0 full, 1 no k loops, 2 k loops alone): /"
done | tee $O/gru_bwd_parts.txt
