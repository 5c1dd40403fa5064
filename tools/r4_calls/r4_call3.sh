#!/bin/bash
# Round 4, call 3: the two-workgroups-per-CU short-K GEMM (parity, micro-bench vs the old dispatch and torch.matmul, step A/B),
# the full-size graph-encoder oracle leg, the measured maxima behind the bf16 bars.
O=gpurun_out/r4c; mkdir -p $O
export PYTHONPATH=$PWD
timeout 300 python -m pytest tests/test_hip_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "short_k or test_gemm" -x > $O/gemm_tests.log 2>&1; tail -5 $O/gemm_tests.log
timeout 300 python -m pytest tests/test_full_size_c2.py -m gpu -q --tb=short -p no:cacheprovider -k "graph_encoder" -s > $O/full_size_encoder.log 2>&1; grep -a "C2 full-size\|passed\|failed\|Error" $O/full_size_encoder.log | head
GTOS_GRAD_TABLE=$O/bf16_grad_table.tsv timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "slice or fused or relation_encoder" -s > $O/measured.log 2>&1; grep -a "MEASURED\|global rel\|worst rel\|passed\|failed" $O/measured.log | cut -c1-400 | head -40
for p2 in 1 0; do
  GTOS_GEMM_P2=$p2 timeout 120 python tools/bench_gemm.py --torch --reps 10 --only "rel_proj" > $O/gemm_p2_$p2.txt 2>&1
  GTOS_GEMM_P2=$p2 timeout 120 python tools/bench_gemm.py --reps 10 --only "gru_tables" >> $O/gemm_p2_$p2.txt 2>&1
  GTOS_GEMM_P2=$p2 timeout 120 python tools/bench_gemm.py --reps 10 --only "relenc_out" >> $O/gemm_p2_$p2.txt 2>&1
  GTOS_GEMM_P2=$p2 timeout 120 python tools/bench_gemm.py --reps 10 --only "gru_hg" >> $O/gemm_p2_$p2.txt 2>&1
  GTOS_GEMM_P2=$p2 timeout 120 python tools/bench_gemm.py --reps 10 --only "ksweep" >> $O/gemm_p2_$p2.txt 2>&1
  echo "== GTOS_GEMM_P2=$p2"; grep -v amdgpu.ids $O/gemm_p2_$p2.txt
done
for leg in p2_1 p2_0 p2_1b; do
  v=1; [ $leg = p2_0 ] && v=0
  GTOS_GEMM_P2=$v GTOS_BENCH_NO_DETAIL=1 timeout 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --prewarm-seconds 10 > $O/bench_$leg.json 2> $O/bench_$leg.err
  python - <<P
import json
try:
    d = json.loads(open("$O/bench_$leg.json").read().strip().splitlines()[-1])
    print("$leg", round(d["ms_per_step"], 2), "ms", d["components"])
except Exception as e:
    print("$leg failed", e); print(open("$O/bench_$leg.err").read()[-1500:])
P
done
