#!/bin/bash
# Round 5, twelfth GPU call: are the forward step's phases bound by INDEPENDENT resources?  Its k loop (or only the k loop's DMA, or only
# its LDS reads + MFMAs) and its cell, each as its own launch, one after the other and side by side on two streams.
O=gpurun_out/r5l; mkdir -p $O
export PYTHONPATH=$PWD
timeout 300 python tools/bench_gru_step.py --concurrent --reps 8 2>&1 | grep -v amdgpu.ids | tee $O/gru_fwd_concurrent.txt
