#!/bin/bash
# Round 5, third GPU call: where the packed path's time is -- isolated kernel timings, SQ / memory counters of the forward and backward
# step kernels, a kernel trace of the default training step, and the default bench line with its detail pass.
O=gpurun_out/r5c; mkdir -p $O
export PYTHONPATH=$PWD
R=$PWD
timeout 300 python tools/bench_gru_step.py --reps 8 > $O/gru_step_isolated.txt 2>&1; cat $O/gru_step_isolated.txt | grep -v amdgpu.ids
GTOS_GRU_FWD_RING=0 timeout 200 python tools/bench_gru_step.py --reps 8 --only fwd 2>&1 | grep -v amdgpu.ids | sed 's/^/single-stage: /' | tee -a $O/gru_step_isolated.txt
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" "FETCH_SIZE" "WRITE_SIZE" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 200 rocprofv3 --kernel-trace --pmc $set -d $R/$O/pmc_$tag -o p -- python $R/tools/bench_gru_step.py --reps 3 --only "fwd bwd" > $R/$O/pmc_$tag.log 2>&1
  echo "pmc $tag rc=$?"
done
cd $R
: > $O/pmc_summary.txt
for d in $O/pmc_*/; do DB=$(find $d -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_pmc.py $DB gru_step >> $O/pmc_summary.txt 2>&1; rm -rf $d; done
cat $O/pmc_summary.txt
cd /tmp && GTOS_BENCH_NO_DETAIL=1 timeout 300 rocprofv3 --kernel-trace -d $R/$O/prof -o trace -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-masks-leg --no-loader-leg --prewarm-seconds 3 > $R/$O/bench_line_under_rocprof.json 2> $R/$O/bench_rocprof.err
cd $R
DB=$(find $O/prof -name "*.db" | head -1)
python tools/rocpd_stats.py $DB $O/kernel_stats.csv > /dev/null
python tools/rocpd_stats.py $DB $O/kernel_stats_by_grid.csv --by-grid > /dev/null
python tools/rocpd_sequence.py $DB --step 3 > $O/step_sequence.txt; head -1 $O/step_sequence.txt
python tools/rocpd_timeline.py $DB > $O/timeline.txt 2>&1 || true
rm -rf $O/prof
head -24 $O/kernel_stats.csv | cut -c1-160
timeout 400 python bench.py --no-cpu-baseline --no-loader-leg --no-masks-leg > $O/bench_default_detail.json 2> $O/bench_default_detail.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5c/bench_default_detail.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], d['components'])
for r in d['roofline']['kernels']: print(r['kernel'][:90], r['launches_per_step'], r['avg_us'], r['achieved'], r['unit'], r['frac'], r['ms_per_step'])
print(d['roofline']['frac'], d['roofline']['in_step'])
PY
