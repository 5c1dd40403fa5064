#!/bin/bash
# Round-3 GPU call 2: the tests touched since call 1, bench, loader-in-the-loop with worker processes, one-step kernel sequence.
O=gpurun_out/r3b; mkdir -p $O
export PYTHONPATH=$PWD
( time timeout 600 python -m pytest tests/test_full_size_c2.py -m gpu -q -x --tb=short -p no:cacheprovider -s > $O/full_size.log 2>&1 ) 2> $O/full_size.time
tail -3 $O/full_size.log; cat $O/full_size.time | grep real
( time timeout 900 python -m pytest tests/test_hip_parity.py tests/test_beam_and_vocab.py -m gpu -q --tb=short -p no:cacheprovider \
   -k "any_head_geometry or rejects or decode_step or trie or relation_encoder or gru or step_control or bench or beam_search or incremental or generator_vs_golden or c2_slice or factored or edge_shapes or dropout" \
   > $O/changed_tests.log 2>&1 ) 2> $O/changed_tests.time
tail -15 $O/changed_tests.log; grep real $O/changed_tests.time
timeout 240 python bench.py --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err; python - <<P
import json
try:
    d=json.load(open("$O/bench_c2.json")); print("bench", d["value"], d["ms_per_step"], d["components"])
except Exception as e: print("bench failed", e); print(open("$O/bench_c2.err").read()[-2000:])
P
for w in 4 6; do
  GTOS_BENCH_NO_DETAIL=1 timeout 200 python bench.py --no-cpu-baseline --fresh-batches --workers $w --steps 20 > $O/bench_c2_fresh_p$w.json 2> $O/bench_c2_fresh_p$w.err
  python - <<P
import json
try:
    d=json.load(open("$O/bench_c2_fresh_p$w.json")); print("fresh procs=$w", d["value"], d["ms_per_step"], d["config"]["loader"], d["components"])
except Exception as e: print("fresh procs=$w failed", e); print(open("$O/bench_c2_fresh_p$w.err").read()[-1500:])
P
done
cd /tmp && export TMPDIR=/tmp && GTOS_BENCH_NO_DETAIL=1 timeout 300 rocprofv3 --kernel-trace -d $OLDPWD/$O/prof -o trace -- python $OLDPWD/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $OLDPWD/$O/bench_rocprof.json 2> $OLDPWD/$O/bench_rocprof.err
cd $OLDPWD
DB=$(find $O/prof -name "*.db" | head -1); echo "db $DB"
python tools/rocpd_stats.py $DB $O/kernel_stats.csv > /dev/null && head -25 $O/kernel_stats.csv
python tools/rocpd_sequence.py $DB --step 3 > $O/step_sequence.txt; head -3 $O/step_sequence.txt
rm -rf $O/prof
