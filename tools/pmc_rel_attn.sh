#!/bin/bash
# HBM traffic of the relation-attention kernels from rocprofv3 PMC counters, PER OPERAND MODE (dense signature / factored),
# FETCH_SIZE and WRITE_SIZE in SEPARATE passes (they do not fit one pass, MI355X_MICROARCH.md "rocprofv3 PMC slots"), no
# tracing domain other than --kernel-trace.  Run on the GPU box from the repo root:
#     bash tools/pmc_rel_attn.sh          -> gpurun_out/${ROUND:-r3}_rel_attn_pmc.json (+ the per-kernel counter table .txt)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
OUT=gpurun_out/pmc
rm -rf $OUT; mkdir -p $OUT
for mode in dense factored; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $ctr -d $OUT/${mode}_${ctr} -o p -- \
        python tools/bench_rel_attn.py --mode $mode --reps 4 > $OUT/${mode}_${ctr}.log 2>&1
  done
done
python tools/pmc_summary.py $OUT gpurun_out/${ROUND:-r3}_rel_attn_pmc.json > gpurun_out/${ROUND:-r3}_rel_attn_pmc_counters.txt
cat gpurun_out/${ROUND:-r3}_rel_attn_pmc_counters.txt
rm -rf $OUT
