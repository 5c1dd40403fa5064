#!/bin/bash
# non-temporal hint on both (streamed-once) operands of the TN weight-gradient kernel
export PYTHONPATH=$PWD
for nt in 0 1 0 1; do
  echo "== GTOS_GEMM_TN_NT=$nt"
  GTOS_GEMM_TN_NT=$nt timeout 120 python tools/bench_gemm.py --reps 5 2>&1 | grep -v amdgpu.ids | grep " TN " | cut -c1-100
done
for nt in 0 1 0 1; do
  GTOS_GEMM_TN_NT=$nt GTOS_BENCH_NO_DETAIL=1 timeout 200 python bench.py --no-cpu-baseline --no-loader-leg --no-masks-leg --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('TN_NT=$nt', round(d['ms_per_step'],2),'ms', round(d['value'],1))"
done
