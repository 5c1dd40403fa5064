#!/bin/bash
# Round 4, call 7: mode-0 attention specialisation (decoder launches): micro-benchmark and step A/B, the decoder-side parity tests.
O=gpurun_out/r4g; mkdir -p $O
export PYTHONPATH=$PWD
for v in 1 0; do echo "== GTOS_ATTN_M0=$v"; GTOS_ATTN_M0=$v timeout 200 python tools/bench_attn_mode0.py 2>&1 | grep -v amdgpu.ids | tee $O/attn_mode0_$v.txt; done
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "transformer_layer or generator or attention or attn or slice or beam or decode" > $O/tests.log 2>&1; tail -4 $O/tests.log
for leg in m0_1 m0_0 m0_1b m0_0b; do
  v=1; case $leg in m0_0*) v=0;; esac
  GTOS_ATTN_M0=$v GTOS_BENCH_NO_DETAIL=1 timeout 300 python bench.py --no-cpu-baseline --no-masks-leg --no-loader-leg --steps 20 --warmup 5 --prewarm-seconds 10 > $O/bench_$leg.json 2> $O/bench_$leg.err
  python - <<P
import json
try:
    d = json.loads(open("$O/bench_$leg.json").read().strip().splitlines()[-1])
    print("$leg", round(d["ms_per_step"], 2), "ms")
except Exception as e:
    print("$leg failed", e); print(open("$O/bench_$leg.err").read()[-2500:])
P
done
for v in 1 0; do GTOS_ATTN_M0=$v GTOS_BENCH_NO_DETAIL=1 timeout 300 python bench.py --config C1 --no-cpu-baseline --no-masks-leg --no-loader-leg --steps 50 --warmup 10 --prewarm-seconds 5 2> /dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C1 m0=$v', round(d['ms_per_step'],2))"; done
