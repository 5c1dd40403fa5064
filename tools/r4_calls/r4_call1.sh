#!/bin/bash
# Round 4, first GPU call: (1) HBM access-pattern ceilings, (2) the device builders' GPU cases + equality / build times at C2,
# (3) loader-in-the-loop legs, (4) the round's starting bench line, (5) GEMM shapes against torch.matmul.
# Usage: gpurun --timeout 600 -- 'bash tools/r4_call1.sh'
O=gpurun_out/r4a; mkdir -p $O
export PYTHONPATH=$PWD
timeout 120 tools/probes/membw_probe > $O/membw_probe.txt 2>&1; tail -60 $O/membw_probe.txt
timeout 200 python -m pytest tests/test_zz_hip_trie_builder.py tests/zzz_hip_relbatch_cases.py -m gpu -q --tb=short -p no:cacheprovider > $O/gpu_tests_builders.log 2>&1
tail -5 $O/gpu_tests_builders.log
timeout 90 python tools/hip_relbatch_check.py $O/hip_relbatch_check.json 2> $O/hip_relbatch_check.err | cut -c1-1500
GTOS_BENCH_NO_DETAIL=0 timeout 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_start.json 2> $O/bench_start.err
python - <<P
import json
try:
    d = json.loads(open("$O/bench_start.json").read().strip().splitlines()[-1])
    print("start", round(d["ms_per_step"], 2), "ms", d.get("components"), d["roofline"]["frac"], d["roofline"]["in_step"]["frac"])
except Exception as e:
    print("bench failed", e); print(open("$O/bench_start.err").read()[-1500:])
P
B="python bench.py --fresh-batches --no-cpu-baseline --steps 30 --warmup 3"
leg() {  # name, args...
  n=$1; shift
  GTOS_BENCH_NO_DETAIL=1 timeout 90 $B "$@" > $O/bench_$n.json 2> $O/bench_$n.err
  python - <<P
import json
try:
    d = json.loads(open("$O/bench_$n.json").read().strip().splitlines()[-1]); l = d["config"]["loader"]
    print("$n", round(d["ms_per_step"], 2), "ms", round(d["value"], 1), "graphs/s wait", l["consumer_wait_ms_per_step"], "asm", l["host_assembly_s_per_batch"])
except Exception as e:
    print("$n failed", e); print(open("$O/bench_$n.err").read()[-1500:])
P
}
leg host_w4 --workers 4 --prewarm-seconds 3
leg hiptries_w1 --workers 1 --device-tries hip --prewarm-seconds 3
leg devrel_w1 --workers 1 --device-relations --prewarm-seconds 3
leg devrel_w1_threads --workers 1 --loader threads --device-relations --prewarm-seconds 3
leg devrel_w1_prepworker --workers 1 --device-relations --prep-in-worker --prewarm-seconds 3
timeout 150 python tools/bench_gemm.py --torch --reps 5 > $O/gemm_vs_torch.txt 2>&1; grep -v amdgpu.ids $O/gemm_vs_torch.txt | head -80
