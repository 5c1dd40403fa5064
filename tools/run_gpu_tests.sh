#!/bin/bash
# Runs every gpu-marked test function in its own process (a GPU memory fault then only kills that one) and
# collects the logs under gpurun_out/.  Usage on the GPU box: bash tools/run_gpu_tests.sh
mkdir -p gpurun_out
export PYTHONPATH=$PWD
: > gpurun_out/gpu_tests_summary.txt
for t in $(python -m pytest tests -m gpu --collect-only -q 2>/dev/null | grep '::' | sed 's/\[.*//' | sort -u); do
  name=$(echo $t | sed 's/.*:://')
  timeout 600 python -m pytest "$t" -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/test_$name.log 2>&1
  rc=$?
  echo "$name rc=$rc $(tail -1 gpurun_out/test_$name.log)" | tee -a gpurun_out/gpu_tests_summary.txt
done
