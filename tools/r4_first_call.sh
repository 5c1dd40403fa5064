#!/bin/bash
# First GPU call of round 4 (~3 GPU-minutes): what round 3 wrote after its GPU minutes were spent.
#   1. the GPU tests of the device-side batch assembly (HIP builders of the tries / relation tensors / relation index), no -x
#   2. tools/hip_relbatch_check.py: equality with the host builders and build times at C2 size
#   3. loader in the loop: host tries vs HIP tries vs everything on the device, 1 and 2 worker processes, after a common warm-up leg
# Usage: gpurun --timeout 420 -- 'bash tools/r4_first_call.sh'
O=gpurun_out/r4a; mkdir -p $O
export PYTHONPATH=$PWD
timeout 150 python -m pytest tests/test_zz_hip_trie_builder.py tests/zzz_hip_relbatch_cases.py -m gpu -q --tb=short -p no:cacheprovider > $O/gpu_tests_builders.log 2>&1
tail -5 $O/gpu_tests_builders.log
timeout 60 python tools/hip_relbatch_check.py $O/hip_relbatch_check.json 2> $O/hip_relbatch_check.err | cut -c1-1500
B="python bench.py --fresh-batches --no-cpu-baseline --steps 30 --warmup 3"
leg() {  # name, args...
  n=$1; shift
  GTOS_BENCH_NO_DETAIL=1 timeout 60 $B "$@" > $O/bench_$n.json 2> $O/bench_$n.err
  python - <<P
import json
try:
    d = json.loads(open("$O/bench_$n.json").read().strip().splitlines()[-1]); l = d["config"]["loader"]
    print("$n", round(d["ms_per_step"], 2), "ms", round(d["value"], 1), "graphs/s wait", l["consumer_wait_ms_per_step"], "asm", l["host_assembly_s_per_batch"])
except Exception as e:
    print("$n failed", e); print(open("$O/bench_$n.err").read()[-1500:])
P
}
leg warmup_host_w4 --workers 4 --prewarm-seconds 20
leg host_w2 --workers 2 --prewarm-seconds 3
leg hiptries_w2 --workers 2 --device-tries hip --prewarm-seconds 3
leg hiptries_w1 --workers 1 --device-tries hip --prewarm-seconds 3
leg devrel_w1 --workers 1 --device-relations --prewarm-seconds 3
leg devrel_w1_threads --workers 1 --loader threads --device-relations --prewarm-seconds 3
leg devrel_w1_prepworker --workers 1 --device-relations --prep-in-worker --prewarm-seconds 3
leg hiptries_w2_prepworker --workers 2 --device-tries hip --prep-in-worker --prewarm-seconds 3
leg host_w4_again --workers 4 --prewarm-seconds 3
