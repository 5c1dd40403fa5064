#!/bin/bash
# Round-3 GPU call 4: same-box A/B of the switches (children-sum indirection, layer-1 table overlap, projection prefetch),
# embed backward, dropout A/B with a sane learning rate.
O=gpurun_out/r3d; mkdir -p $O
export PYTHONPATH=$PWD
( time timeout 600 python -m pytest tests/test_hip_parity.py tests/test_full_size_c2.py -m gpu -q --tb=short -p no:cacheprovider \
   -k "trie or relation_encoder or gru or embed or generator_vs_golden or c2_slice or full_bank or large_vocabulary or segment or baseline_config" \
   > $O/tests.log 2>&1 ) 2> $O/tests.time
tail -6 $O/tests.log; grep real $O/tests.time
run() { # name, env...
  name=$1; shift
  env "$@" GTOS_BENCH_NO_DETAIL=1 timeout 200 python bench.py --no-cpu-baseline --steps 12 > $O/bench_$name.json 2> $O/bench_$name.err
  python - <<P
import json
try:
    d=json.load(open("$O/bench_$name.json")); print("bench %-14s" % "$name", round(d["value"],1), round(d["ms_per_step"],3), {k: round(v,2) for k,v in d["components"].items() if k != "note"})
except Exception as e: print("bench $name failed", e); print(open("$O/bench_$name.err").read()[-1500:])
P
}
run default A=1
run sumidx0 GTOS_GRU_SUMIDX=0
run l1tables0 GTOS_GRU_L1_TABLES=0
run projside1 GTOS_PROJ_SIDE=1
run default2 A=1
cd /tmp && export TMPDIR=/tmp && GTOS_BENCH_NO_DETAIL=1 timeout 300 rocprofv3 --kernel-trace -d $OLDPWD/$O/prof -o trace -- python $OLDPWD/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $OLDPWD/$O/bench_rocprof.json 2> $OLDPWD/$O/bench_rocprof.err
cd $OLDPWD
DB=$(find $O/prof -name "*.db" | head -1)
python tools/rocpd_stats.py $DB $O/kernel_stats.csv > /dev/null
python tools/rocpd_sequence.py $DB --step 3 > $O/step_sequence.txt; head -1 $O/step_sequence.txt
rm -rf $O/prof
timeout 500 python tools/dropout_ab.py --steps 4000 --train-trees 256 --every 200 --warmup 1500 --out $O/dropout_ab_256_w1500.json > $O/dropout_ab_256_w1500.log 2>&1; tail -23 $O/dropout_ab_256_w1500.log
