#!/bin/bash
# Round-3 GPU call 3: children-sum indirection, chunked bank gradient, embed backward layout; loader depth; over-fitting A/B.
O=gpurun_out/r3c; mkdir -p $O
export PYTHONPATH=$PWD
( time timeout 600 python -m pytest tests/test_hip_parity.py tests/test_full_size_c2.py -m gpu -q --tb=short -p no:cacheprovider \
   -k "trie or relation_encoder or gru or batched_bank or embed or side_stream or generator_vs_golden or c2_slice or strong or full_batch or full_bank or large_vocabulary or segment" \
   > $O/tests.log 2>&1 ) 2> $O/tests.time
tail -12 $O/tests.log; grep real $O/tests.time
for v in 2 0; do
GTOS_DX_CHUNK=$v timeout 240 python bench.py --no-cpu-baseline > $O/bench_c2_dx$v.json 2> $O/bench_c2_dx$v.err; python - <<P
import json
try:
    d=json.load(open("$O/bench_c2_dx$v.json")); print("bench dxchunk=$v", d["value"], d["ms_per_step"], d["components"])
except Exception as e: print("bench failed", e); print(open("$O/bench_c2_dx$v.err").read()[-2000:])
P
done
for w in 4 8; do
  GTOS_BENCH_NO_DETAIL=1 timeout 200 python bench.py --no-cpu-baseline --fresh-batches --workers $w --steps 20 > $O/bench_c2_fresh_p$w.json 2> $O/bench_c2_fresh_p$w.err
  python - <<P
import json
try:
    d=json.load(open("$O/bench_c2_fresh_p$w.json")); print("fresh procs=$w", d["value"], d["ms_per_step"], d["config"]["loader"])
except Exception as e: print("fresh procs=$w failed", e); print(open("$O/bench_c2_fresh_p$w.err").read()[-1500:])
P
done
cd /tmp && export TMPDIR=/tmp && GTOS_BENCH_NO_DETAIL=1 timeout 300 rocprofv3 --kernel-trace -d $OLDPWD/$O/prof -o trace -- python $OLDPWD/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $OLDPWD/$O/bench_rocprof.json 2> $OLDPWD/$O/bench_rocprof.err
cd $OLDPWD
DB=$(find $O/prof -name "*.db" | head -1)
python tools/rocpd_stats.py $DB $O/kernel_stats.csv > /dev/null
python tools/rocpd_sequence.py $DB --step 3 > $O/step_sequence.txt; head -1 $O/step_sequence.txt
rm -rf $O/prof
timeout 400 python tools/dropout_ab.py --steps 3000 --train-trees 256 --every 150 --out $O/dropout_ab_256.json > $O/dropout_ab_256.log 2>&1; tail -24 $O/dropout_ab_256.log
