#!/bin/bash
# Round-4 evidence run (closing): full gpu suite (-s: the measured maxima behind the bf16 bars), smoke, default bench (with cpu_baseline
# and the secondary legs), the other BASELINE configs, the GEMM shapes against torch.matmul (hipBLASLt) on the final code.
O=gpurun_out/r4zz; mkdir -p $O
export PYTHONPATH=$PWD
( time timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -s > $O/gpu_tests.log 2>&1 ) 2> $O/gpu_tests.time
tail -4 $O/gpu_tests.log | cut -c1-300; grep real $O/gpu_tests.time; grep -a "MEASURED\|bf16 vs golden\|C2 full-size\|C2 B=64" $O/gpu_tests.log | cut -c1-260 > $O/measured.txt; wc -l $O/measured.txt
timeout 300 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log | cut -c1-400
( time timeout 600 python bench.py > $O/bench_c2_n1.json 2> $O/bench_c2_n1.err ) 2> $O/bench_c2_n1.time; python - <<P
import json
try:
    d=json.loads(open("$O/bench_c2_n1.json").read().strip().splitlines()[-1]); print("bench", round(d["value"],1), round(d["ms_per_step"],2), d["components"]); print(d["roofline"]["frac"], d["roofline"]["in_step"]["frac"], d["cpu_baseline"]["value"], (d.get("node_masks") or d.get("reference_masks"))["ms_per_step"], d["loader_in_loop"]["ms_per_step"], d["config"]["device_memory"])
    for r in d["roofline"]["kernels"]: print("   ", r["kernel"][:60], r["avg_us"], r["frac"])
except Exception as e: print("bench failed", e); print(open("$O/bench_c2_n1.err").read()[-2000:])
P
grep real $O/bench_c2_n1.time
for c in C1 C3 C5; do
  GTOS_BENCH_NO_DETAIL=1 timeout 400 python bench.py --config $c --steps 8 --warmup 3 --no-cpu-baseline --prewarm-seconds 8 > $O/bench_$c.json 2> $O/bench_$c.err
  python -c "
import json
try:
    d=json.loads(open('$O/bench_$c.json').read().strip().splitlines()[-1]); print('$c', round(d['value'],1), round(d['ms_per_step'],2), d['roofline']['frac'], d['config']['device_memory'], (d.get('node_masks') or d.get('reference_masks') or {}).get('ms_per_step'), (d.get('loader_in_loop') or {}).get('ms_per_step'))
except Exception as e: print('$c failed', e)"
done
timeout 200 python tools/bench_gemm.py --torch --reps 5 > $O/gemm_vs_hipblaslt.txt 2>&1; grep -v amdgpu.ids $O/gemm_vs_hipblaslt.txt | grep -c "TF/s"
