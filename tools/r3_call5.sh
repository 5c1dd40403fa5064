#!/bin/bash
# Round-3 GPU call 5: fused token-encoder kernels + two-level embed backward (tests), bench with the projection prefetch on,
# refreshed relation-attention PMC traffic, the other BASELINE configs and the decode benchmark.
O=gpurun_out/r3e; mkdir -p $O
export PYTHONPATH=$PWD
( time timeout 600 python -m pytest tests/test_hip_parity.py tests/test_beam_and_vocab.py -m gpu -q --tb=short -p no:cacheprovider \
   -k "token_encoder or embed or generator_vs_golden or c2_slice or c3_slice or trie or relation_encoder or large_vocabulary or baseline_config or beam_search or golden or trainer" \
   > $O/tests.log 2>&1 ) 2> $O/tests.time
tail -6 $O/tests.log; grep real $O/tests.time
timeout 300 python bench.py --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err; python - <<P
import json
try:
    d=json.load(open("$O/bench_c2.json")); print("bench", d["value"], d["ms_per_step"], d["components"]); print(d["roofline"]["frac"], d["roofline"]["in_step"])
except Exception as e: print("bench failed", e); print(open("$O/bench_c2.err").read()[-2000:])
P
GTOS_BENCH_NO_DETAIL=1 GTOS_PROJ_SIDE=0 timeout 200 python bench.py --no-cpu-baseline --steps 12 > $O/bench_c2_proj0.json 2> $O/bench_c2_proj0.err
python -c "
import json; d=json.load(open('$O/bench_c2_proj0.json')); print('proj0', d['value'], d['ms_per_step'])"
cd /tmp && export TMPDIR=/tmp && GTOS_BENCH_NO_DETAIL=1 timeout 300 rocprofv3 --kernel-trace -d $OLDPWD/$O/prof -o trace -- python $OLDPWD/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $OLDPWD/$O/bench_rocprof.json 2> $OLDPWD/$O/bench_rocprof.err
cd $OLDPWD
DB=$(find $O/prof -name "*.db" | head -1)
python tools/rocpd_stats.py $DB $O/kernel_stats.csv > /dev/null
python tools/rocpd_stats.py $DB $O/kernel_stats_by_grid.csv --by-grid > /dev/null
python tools/rocpd_sequence.py $DB --step 3 > $O/step_sequence.txt; head -1 $O/step_sequence.txt
python tools/rocpd_timeline.py $DB 8 > $O/timeline.txt 2>&1; head -2 $O/timeline.txt
rm -rf $O/prof
ROUND=r3 timeout 600 bash tools/pmc_rel_attn.sh > $O/pmc.log 2>&1; tail -12 $O/pmc.log
for c in C1 C3 C5; do
  GTOS_BENCH_NO_DETAIL=1 timeout 300 python bench.py --config $c --steps 4 --warmup 2 --no-cpu-baseline > $O/bench_$c.json 2> $O/bench_$c.err
  python -c "
import json
try:
    d=json.load(open('$O/bench_$c.json')); print('$c', round(d['value'],1), round(d['ms_per_step'],2), d['roofline']['frac'])
except Exception as e: print('$c failed', e)"
done
timeout 300 python bench.py --decode --no-cpu-baseline > $O/decode_c2.json 2> $O/decode_c2.err; tail -c 400 $O/decode_c2.json
