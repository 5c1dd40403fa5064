#!/bin/bash
# Round 5, call 19: mixed roles in ONE launch of the forward step (measuring switch 7: half the workgroups run the k loop's DMA alone, half
# the cell alone; two workgroups per CU with four waves, neighbouring CUs with eight): do the two phases queue on one resource?
O=gpurun_out/r5s; mkdir -p $O
export PYTHONPATH=$PWD
timeout 300 python tools/bench_gru_step.py --dbuf --reps 8 2>&1 | grep -v amdgpu.ids | tee $O/gru_fwd_mixed_roles.txt
