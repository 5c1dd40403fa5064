#!/bin/bash
# Round 4, call 9: C5 under allocator settings (reserved memory), 8 steps each.
O=gpurun_out/r4i; mkdir -p $O
export PYTHONPATH=$PWD
run() {  # name, alloc conf
  PYTORCH_CUDA_ALLOC_CONF="$2" PYTORCH_ALLOC_CONF="$2" GTOS_BENCH_NO_ROUNDUP=1 GTOS_BENCH_NO_DETAIL=1 timeout 400 python bench.py --config C5 --no-cpu-baseline --no-masks-leg --no-loader-leg --steps 8 --warmup 3 --prewarm-seconds 8 > $O/bench_C5_$1.json 2> $O/bench_C5_$1.err
  python - <<P
import json
try:
    d = json.loads(open("$O/bench_C5_$1.json").read().strip().splitlines()[-1])
    print("C5 $1 [$2]", round(d["ms_per_step"], 1), "ms", d["config"]["device_memory"])
except Exception as e:
    print("C5 $1 failed", e); print(open("$O/bench_C5_$1.err").read()[-800:])
P
}
run default ""
run gc30 "garbage_collection_threshold:0.3"
run gc20 "garbage_collection_threshold:0.2"
run exp "expandable_segments:True"
run expgc "expandable_segments:True,garbage_collection_threshold:0.3"
run split "max_split_size_mb:512"
run splitgc "max_split_size_mb:512,garbage_collection_threshold:0.3"
