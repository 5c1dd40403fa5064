#!/bin/bash
# What bounds the forward-shaped GEMMs?  rocprofv3 PMC counters over the K sweep of tools/bench_gemm.py (M = 434624,
# N = 1024, K = 32 ... 2016), one counter group per pass (a group with an unknown counter name fails alone), no tracing
# domain other than --kernel-trace.  Run on the GPU box from the repo root:
#     bash tools/pmc_gemm.sh          -> gpurun_out/pmc_gemm/<group>.tsv (one row per launch and counter)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
OUT=gpurun_out/pmc_gemm
rm -rf $OUT; mkdir -p $OUT
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum" \
           "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $grp -d $OUT/g$i -o p -- python tools/bench_gemm.py --reps 3 --only ksweep > $OUT/g$i.log 2>&1
  python tools/pmc_dump.py $OUT/g$i > $OUT/g$i.tsv 2>> $OUT/g$i.log
  rm -rf $OUT/g$i
done
wc -l $OUT/*.tsv
