#!/bin/bash
# Round 5, call 13: does a PHASE OFFSET between the two workgroups of a CU (four-wave forward step) let one's k loop run under the other's
# cell?  Measuring switches 5 / 6: some first-generation workgroups start late.
O=gpurun_out/r5m; mkdir -p $O
export PYTHONPATH=$PWD
timeout 300 python tools/bench_gru_step.py --phase --reps 8 2>&1 | grep -v amdgpu.ids | tee $O/gru_fwd_phase_offset.txt
