#!/bin/bash
# Round 5, eleventh GPU call: the forward step's k loop split further -- its DMA alone (3) and its fragment reads + MFMAs alone (4).
O=gpurun_out/r5k; mkdir -p $O
export PYTHONPATH=$PWD
for dbg in 0 2 3 4; do GTOS_GRU_DBG=$dbg timeout 200 python tools/bench_gru_step.py --reps 8 --only fwd 2>&1 | grep -v amdgpu.ids | sed "s/^/dbg=$dbg (0 full, 2 k loop, 3 its DMA only, 4 its reads+MFMA only): /" | tee -a $O/gru_fwd_kloop_parts.txt; done
