#!/bin/bash
# gemm_w4_nt_kernel: where does the step time go?  DBG 1 = no global loads, 2 = no LDS traffic either (MFMAs + barriers only),
# 4 = every stage loads the k range of stage 0 (the same lines every step: L1 hits), 5 = the k ranges of stages 0..7 in rotation (L2 hits)
O=gpurun_out/r4w; mkdir -p $O
export PYTHONPATH=$PWD
for d in 0 4 0 1; do
  echo "== DBG=$d"
  GTOS_GEMM_W4=1 GTOS_GEMM_W4_DBG=$d timeout 100 python tools/bench_gemm.py --reps 5 --only deepK_dbank 2>&1 | grep -v amdgpu.ids
  GTOS_GEMM_W4=1 GTOS_GEMM_W4_DBG=$d timeout 100 python tools/bench_gemm.py --reps 5 --only square8k 2>&1 | grep -v amdgpu.ids
done
