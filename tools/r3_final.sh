#!/bin/bash
# Round-3 evidence run: full gpu suite, smoke, default bench (with cpu_baseline), kernel trace, loader-in-the-loop, whole-step
# PMC traffic, the other BASELINE configs, decode.  Everything under gpurun_out/r3z/; the summaries are copied to profiles/.
O=gpurun_out/r3z; mkdir -p $O
export PYTHONPATH=$PWD
( time timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -s > $O/gpu_tests.log 2>&1 ) 2> $O/gpu_tests.time
tail -4 $O/gpu_tests.log; grep real $O/gpu_tests.time; grep "bf16 vs golden" $O/gpu_tests.log
timeout 300 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 400 python bench.py > $O/bench_c2_n1.json 2> $O/bench_c2_n1.err; python - <<P
import json
try:
    d=json.load(open("$O/bench_c2_n1.json")); print("bench", d["value"], d["ms_per_step"], d["components"]); print(d["roofline"]["frac"], d["roofline"]["avg_us"], d["cpu_baseline"]["value"], d["cpu_baseline"]["sample"][:80])
except Exception as e: print("bench failed", e); print(open("$O/bench_c2_n1.err").read()[-2000:])
P
cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace -d $OLDPWD/$O/prof -o trace -- python $OLDPWD/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $OLDPWD/$O/bench_line_under_rocprof.json 2> $OLDPWD/$O/bench_rocprof.err
cd $OLDPWD
DB=$(find $O/prof -name "*.db" | head -1)
python tools/rocpd_stats.py $DB $O/kernel_stats.csv > /dev/null
python tools/rocpd_stats.py $DB $O/kernel_stats_by_grid.csv --by-grid > /dev/null
python tools/rocpd_sequence.py $DB --step 3 > $O/step_sequence.txt; head -1 $O/step_sequence.txt
python tools/rocpd_timeline.py $DB 8 > $O/timeline.txt 2>&1; head -1 $O/timeline.txt
rm -rf $O/prof
for w in 2 4; do
  GTOS_BENCH_NO_DETAIL=1 timeout 200 python bench.py --no-cpu-baseline --fresh-batches --workers $w --steps 30 > $O/bench_c2_fresh_p$w.json 2> $O/bench_c2_fresh_p$w.err
  python - <<P
import json
try:
    d=json.load(open("$O/bench_c2_fresh_p$w.json")); print("fresh procs=$w", d["value"], d["ms_per_step"], d["config"]["loader"])
except Exception as e: print("fresh procs=$w failed", e); print(open("$O/bench_c2_fresh_p$w.err").read()[-1500:])
P
done
GTOS_BENCH_NO_DETAIL=1 timeout 200 python bench.py --no-cpu-baseline --steps 30 > $O/bench_c2_prebuilt_30.json 2> /dev/null; python -c "
import json; d=json.load(open('$O/bench_c2_prebuilt_30.json')); print('prebuilt 30 steps', d['value'], d['ms_per_step'])"
for c in C1 C3 C5; do
  GTOS_BENCH_NO_DETAIL=1 timeout 300 python bench.py --config $c --steps 6 --warmup 3 --no-cpu-baseline > $O/bench_$c.json 2> $O/bench_$c.err
  python -c "
import json
try:
    d=json.load(open('$O/bench_$c.json')); print('$c', round(d['value'],1), round(d['ms_per_step'],2), d['roofline']['frac'])
except Exception as e: print('$c failed', e)"
done
timeout 300 python bench.py --decode --no-cpu-baseline > $O/decode_c2.json 2> $O/decode_c2.err; tail -c 300 $O/decode_c2.json; echo
timeout 600 bash tools/pmc_step.sh > $O/pmc_step.log 2>&1; tail -5 $O/pmc_step.log
