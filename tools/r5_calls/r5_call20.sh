#!/bin/bash
# Round 5, call 20: (a) the forward step kernels' bit identity as a test (five kernel forms, child processes); (b) TIMING of the cell with
# line-wise accesses (measuring switches 8 / 9: 8 consecutive lanes per 128-byte row segment; values in the wrong places) and mixed roles again
O=gpurun_out/r5t; mkdir -p $O
export PYTHONPATH=$PWD
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "ring_kernel_bit_identical or packed_path or fused_step" -p no:cacheprovider 2>&1 | tail -5 | tee $O/tests.txt
timeout 300 python tools/bench_gru_step.py --dbuf --reps 8 2>&1 | grep -v amdgpu.ids | tee $O/gru_fwd_linewise.txt
