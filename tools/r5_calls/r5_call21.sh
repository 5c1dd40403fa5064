#!/bin/bash
# Round 5, call 21: backward step launch with 256-column input-gradient tiles (GTOS_GRU_BWD_WIDE, default 1) and at two workgroups per CU
# (GTOS_GRU_BWD_DBG=3: no spills with the wide tiles) against the first version; parity of the role first.
O=gpurun_out/r5u; mkdir -p $O
export PYTHONPATH=$PWD
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "input_gradient_role or packed_path or fused_step" -p no:cacheprovider 2>&1 | tail -3 | tee $O/tests.txt
for w in 0 1; do for d in 0 3; do
  GTOS_GRU_BWD_WIDE=$w GTOS_GRU_BWD_DBG=$d timeout 300 python tools/bench_gru_step.py --only bwd,dinp --reps 8 2>&1 | grep -v amdgpu.ids | sed "s/^/wide=$w occupancy=$d (0: three workgroups per CU, 3: two): /"
done; done | tee $O/gru_bwd_wide.txt
