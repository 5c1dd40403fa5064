#!/bin/bash
# Round 4: gemm_w4_nt_kernel (4 waves x 128x128, deep K) -- correctness, then rate against the 8-wave kernel and hipBLASLt.
O=gpurun_out/r4w; mkdir -p $O
export PYTHONPATH=$PWD
timeout 400 python -m pytest tests/test_hip_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "deep_k_one_wave" > $O/w4_tests.log 2>&1; tail -15 $O/w4_tests.log
for w in 0 1; do
  echo "== GTOS_GEMM_W4=$w"
  GTOS_GEMM_W4=$w timeout 200 python tools/bench_gemm.py --torch --reps 5 --only deepK > $O/gemm_w4_$w.txt 2>&1; grep -v amdgpu.ids $O/gemm_w4_$w.txt
  GTOS_GEMM_W4=$w timeout 100 python tools/bench_gemm.py --reps 5 --only square8k 2>&1 | grep -v amdgpu.ids
  GTOS_GEMM_W4=$w GTOS_GEMM_W4_MINK=1024 timeout 100 python tools/bench_gemm.py --reps 5 --only ksweep1024 2>&1 | grep -v amdgpu.ids
done
