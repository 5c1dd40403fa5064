#!/bin/bash
# Round 4, call 4: role-split deep-ring short-K GEMM (parity, micro-bench, step A/B); bench.py with the loader_in_loop leg; fresh batches
# for C2 (default route) and C3 (DependencyLoader); the new loader-default GPU test.
O=gpurun_out/r4d; mkdir -p $O
export PYTHONPATH=$PWD
timeout 300 python -m pytest tests/test_hip_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "short_k or test_gemm" -x > $O/gemm_tests.log 2>&1; tail -5 $O/gemm_tests.log
for p2 in 1 0; do
  GTOS_GEMM_P2=$p2 timeout 120 python tools/bench_gemm.py --torch --reps 10 --only "rel_proj" > $O/gemm_p2_$p2.txt 2>&1
  for sh in gru_tables relenc_out gru_hg ksweep128 ksweep256 ksweep512; do GTOS_GEMM_P2=$p2 timeout 120 python tools/bench_gemm.py --reps 10 --only "$sh" >> $O/gemm_p2_$p2.txt 2>&1; done
  echo "== GTOS_GEMM_P2=$p2"; grep -v amdgpu.ids $O/gemm_p2_$p2.txt
done
timeout 300 python -m pytest tests/test_zzz_hip_relbatch.py tests/test_full_size_c2.py -m gpu -q --tb=short -p no:cacheprovider -k "loader_default or graph_encoder" -s > $O/new_tests.log 2>&1; grep -a "C2 full-size\|passed\|failed\|Error" $O/new_tests.log | head
for leg in p2_1 p2_0; do
  v=1; [ $leg = p2_0 ] && v=0
  GTOS_GEMM_P2=$v GTOS_BENCH_NO_DETAIL=1 timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --prewarm-seconds 10 > $O/bench_$leg.json 2> $O/bench_$leg.err
  python - <<P
import json
try:
    d = json.loads(open("$O/bench_$leg.json").read().strip().splitlines()[-1])
    print("$leg", round(d["ms_per_step"], 2), "ms", d["components"], "\n   loader_in_loop", d.get("loader_in_loop"))
except Exception as e:
    print("$leg failed", e); print(open("$O/bench_$leg.err").read()[-2500:])
P
done
for cfgn in C2 C3; do
  GTOS_BENCH_NO_DETAIL=1 timeout 300 python bench.py --config $cfgn --fresh-batches --no-cpu-baseline --steps 20 --warmup 3 --prewarm-seconds 5 > $O/bench_fresh_$cfgn.json 2> $O/bench_fresh_$cfgn.err
  python - <<P
import json
try:
    d = json.loads(open("$O/bench_fresh_$cfgn.json").read().strip().splitlines()[-1])
    print("fresh $cfgn", round(d["ms_per_step"], 2), "ms", d["config"]["loader"])
except Exception as e:
    print("fresh $cfgn failed", e); print(open("$O/bench_fresh_$cfgn.err").read()[-2500:])
P
done
GTOS_BENCH_NO_DETAIL=1 timeout 300 python bench.py --config C3 --no-cpu-baseline --steps 20 --warmup 3 --prewarm-seconds 5 > $O/bench_C3.json 2> $O/bench_C3.err; python -c "
import json; d=json.loads(open('$O/bench_C3.json').read().strip().splitlines()[-1]); print('C3 prebuilt', round(d['ms_per_step'],2), d.get('loader_in_loop'))" || tail -20 $O/bench_C3.err
