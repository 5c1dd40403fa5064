#!/bin/bash
# gemm_w4_nt_kernel with the non-temporal hint on A: correctness, rates, B hint variant, and the training step with it
O=gpurun_out/r4w; mkdir -p $O
export PYTHONPATH=$PWD
timeout 400 python -m pytest tests/test_hip_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "deep_k_one_wave" > $O/w4_tests.log 2>&1; tail -3 $O/w4_tests.log
for d in 0 9 6; do
  echo "== GTOS_GEMM_W4=1 DBG=$d"
  GTOS_GEMM_W4=1 GTOS_GEMM_W4_DBG=$d timeout 200 python tools/bench_gemm.py --reps 5 --only deepK 2>&1 | grep -v amdgpu.ids
  GTOS_GEMM_W4=1 GTOS_GEMM_W4_DBG=$d timeout 100 python tools/bench_gemm.py --reps 5 --only square8k 2>&1 | grep -v amdgpu.ids
  GTOS_GEMM_W4=1 GTOS_GEMM_W4_DBG=$d GTOS_GEMM_W4_MINK=1024 timeout 100 python tools/bench_gemm.py --reps 5 --only ksweep 2>&1 | grep -v amdgpu.ids | grep "1024\|2016"
done
for w in 0 1 0 1; do
  GTOS_GEMM_W4=$w GTOS_BENCH_NO_DETAIL=1 timeout 200 python bench.py --no-cpu-baseline --no-loader-leg --no-masks-leg --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('W4=$w', round(d['ms_per_step'],2),'ms', round(d['value'],1))"
done
