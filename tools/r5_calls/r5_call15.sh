#!/bin/bash
# Round 5, call 15: whole cache lines per DMA row in the forward step's k loop (two slots of 64-k stages, gru_step_fwd_dbuf_kernel) against
# the three-slot ring of 32-k stages: whole launch, k loop alone, DMA alone, cell alone; bit identity.
O=gpurun_out/r5o; mkdir -p $O
export PYTHONPATH=$PWD
timeout 300 python tools/bench_gru_step.py --dbuf --reps 8 2>&1 | grep -v amdgpu.ids | tee $O/gru_fwd_dbuf.txt
