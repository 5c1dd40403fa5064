#!/bin/bash
# Round-3 GPU call 6: embedding gradient as one-hot GEMMs (tests + same-box A/B), C5 regression hunt.
O=gpurun_out/r3f; mkdir -p $O
export PYTHONPATH=$PWD
( time timeout 600 python -m pytest tests/test_hip_parity.py tests/test_full_size_c2.py -m gpu -q --tb=short -p no:cacheprovider \
   -k "trie or relation_encoder or embed or generator_vs_golden or c2_slice or c3_slice or full_bank or side_stream or baseline_config" \
   > $O/tests.log 2>&1 ) 2> $O/tests.time
tail -6 $O/tests.log; grep real $O/tests.time
run() { name=$1; cfg=$2; steps=$3; shift 3
  env "$@" GTOS_BENCH_NO_DETAIL=1 timeout 300 python bench.py --config $cfg --no-cpu-baseline --steps $steps --warmup 2 > $O/bench_$name.json 2> $O/bench_$name.err
  python - <<P
import json
try:
    d=json.load(open("$O/bench_$name.json")); print("bench %-16s" % "$name", round(d["value"],1), round(d["ms_per_step"],3), {k: round(v,2) for k,v in d["components"].items() if k != "note"})
except Exception as e: print("bench $name failed", e); print(open("$O/bench_$name.err").read()[-1500:])
P
}
run c2_default C2 12 A=1
run c2_embgemm0 C2 12 GTOS_EMBED_GEMM=0
run c2_default2 C2 12 A=1
run c5_default C5 4 A=1
run c5_proj0 C5 4 GTOS_PROJ_SIDE=0
run c5_sumidx0 C5 4 GTOS_GRU_SUMIDX=0
run c5_r2like C5 4 GTOS_PROJ_SIDE=0 GTOS_GRU_SUMIDX=0 GTOS_EMBED_GEMM=0
