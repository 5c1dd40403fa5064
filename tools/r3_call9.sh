#!/bin/bash
O=gpurun_out/r3i; mkdir -p $O
export PYTHONPATH=$PWD
timeout 300 python -m pytest tests/test_hip_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "bf16_vs_golden or fresh or prefetcher" 2>&1 | tail -3
GTOS_BENCH_NO_DETAIL=1 timeout 200 python bench.py --no-cpu-baseline --steps 30 > $O/bench_c2_prebuilt_a.json 2> /dev/null; python -c "
import json; d=json.load(open('$O/bench_c2_prebuilt_a.json')); print('prebuilt a', round(d['value'],1), round(d['ms_per_step'],2))"
for w in 2 3 4 6; do
  GTOS_BENCH_NO_DETAIL=1 timeout 200 python bench.py --no-cpu-baseline --fresh-batches --workers $w --steps 30 > $O/bench_c2_fresh_p$w.json 2> $O/bench_c2_fresh_p$w.err
  python - <<P
import json
try:
    d=json.load(open("$O/bench_c2_fresh_p$w.json")); print("fresh procs=$w", round(d["value"],1), round(d["ms_per_step"],2), d["config"]["loader"])
except Exception as e: print("fresh procs=$w failed", e); print(open("$O/bench_c2_fresh_p$w.err").read()[-1500:])
P
done
GTOS_BENCH_NO_DETAIL=1 timeout 200 python bench.py --no-cpu-baseline --fresh-batches --workers 4 --relbatch-threads 1 --steps 30 > $O/bench_c2_fresh_p4_t1.json 2> /dev/null; python -c "
import json; d=json.load(open('$O/bench_c2_fresh_p4_t1.json')); print('fresh p4 t1', round(d['value'],1), round(d['ms_per_step'],2), d['config']['loader'])"
GTOS_BENCH_NO_DETAIL=1 timeout 200 python bench.py --no-cpu-baseline --steps 30 > $O/bench_c2_prebuilt_b.json 2> /dev/null; python -c "
import json; d=json.load(open('$O/bench_c2_prebuilt_b.json')); print('prebuilt b', round(d['value'],1), round(d['ms_per_step'],2))"
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; python -c "import os; print(len(os.sched_getaffinity(0)))"
