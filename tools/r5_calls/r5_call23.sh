#!/bin/bash
# Round 5, call 23: the forward k loop's DMA alone with TWO stages (112 KB per CU) in flight instead of one (measuring switch 10)
O=gpurun_out/r5w; mkdir -p $O
export PYTHONPATH=$PWD
timeout 300 python tools/bench_gru_step.py --dbuf --reps 8 --rows 434624 2>&1 | grep -v amdgpu.ids | grep "8 waves" | tee $O/gru_fwd_two_in_flight.txt
