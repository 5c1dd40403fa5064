#!/bin/bash
# HBM traffic (rocprofv3 PMC: FETCH_SIZE, WRITE_SIZE in SEPARATE passes, --kernel-trace only) of the dominant kernels of the C2
# training step (MASKS=path, the library default, or node): two steps of bench.py per pass.  Run on the GPU box from the repo root:
#     bash tools/pmc_step.sh        -> gpurun_out/${ROUND:-r3}_step_pmc.json (+ table .txt)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
OUT=gpurun_out/pmcstep
rm -rf $OUT; mkdir -p $OUT
for ctr in FETCH_SIZE WRITE_SIZE; do
  GTOS_BENCH_NO_DETAIL=1 timeout 400 rocprofv3 --kernel-trace --pmc $ctr -d $OUT/$ctr -o p -- \
      python bench.py --relation-masks ${MASKS:-path} --steps 2 --warmup 1 --no-cpu-baseline --no-masks-leg --no-loader-leg --prewarm-seconds 3 > $OUT/$ctr.log 2>&1
done
python tools/pmc_step_summary.py $OUT gpurun_out/${ROUND:-r3}_step_pmc.json > gpurun_out/${ROUND:-r3}_step_pmc.txt
cat gpurun_out/${ROUND:-r3}_step_pmc.txt
rm -rf $OUT
