#!/bin/bash
O=gpurun_out/r3h; mkdir -p $O
export PYTHONPATH=$PWD
run() { name=$1; cfg=$2; steps=$3; shift 3
  env "$@" GTOS_BENCH_NO_DETAIL=1 timeout 300 python bench.py --config $cfg --no-cpu-baseline --steps $steps --warmup 2 > $O/bench_$name.json 2> $O/bench_$name.err
  python - <<P
import json
try:
    d=json.load(open("$O/bench_$name.json")); print("bench %-16s" % "$name", round(d["value"],1), round(d["ms_per_step"],3), {k: round(v,2) for k,v in d["components"].items() if k != "note"})
except Exception as e: print("bench $name failed", e); print(open("$O/bench_$name.err").read()[-1500:])
P
}
run warm1 C2 12 A=1
run warm2 C2 12 A=1
run warm3 C2 12 A=1
run c2_default_a C2 12 A=1
run c2_pipe512_a C2 12 GTOS_GEMM_PIPE_MINK=512
run c2_default_b C2 12 A=1
run c2_pipe512_b C2 12 GTOS_GEMM_PIPE_MINK=512
run c2_pipe256 C2 12 GTOS_GEMM_PIPE_MINK=256
run c2_default_c C2 12 A=1
GTOS_GEMM_PIPE_MINK=256 timeout 200 python tools/bench_gemm.py --reps 5 --only NT > $O/bench_gemm_nt_pipe256.txt 2>&1; grep -E "rel_proj|relenc_out|gru_tables|ksweep256|ksweep512|gru_hg|gru_xg_l1" $O/bench_gemm_nt_pipe256.txt
