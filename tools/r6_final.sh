#!/bin/bash
# Round-6 evidence run: full gpu suite (-s: the measured maxima behind the bf16 bars), smoke, the default bench line (cpu_baseline and the
# secondary legs), the decode record, the other BASELINE configs in the default mode with their device memory, the kernel trace / timeline of
# the default step, the counter profiles (attention kernels per operand mode; the step by kernel), every GEMM shape against hipBLASLt.
#     gpurun --timeout 2700 -- 'bash tools/r6_final.sh'        (about 14 GPU-minutes; PROTOCOL=1 adds SURVEY 8d's live CPU protocol, +6)
O=gpurun_out/${OUT:-r6zz}; mkdir -p $O
export PYTHONPATH=$PWD
R=$PWD
( time timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -s > $O/gpu_tests.log 2>&1 ) 2> $O/gpu_tests.time
tail -4 $O/gpu_tests.log | cut -c1-300; grep real $O/gpu_tests.time; grep -a "MEASURED\|bf16 vs golden\|C2 full\|C2 B=64" $O/gpu_tests.log | cut -c1-300 > $O/measured.txt; wc -l $O/measured.txt
timeout 300 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log | cut -c1-400
( time timeout 600 python bench.py > $O/bench_c2_n1.json 2> $O/bench_c2_n1.err ) 2> $O/bench_c2_n1.time; python - <<P
import json
try:
    d=json.loads(open("$O/bench_c2_n1.json").read().strip().splitlines()[-1]); print("bench", round(d["value"],1), round(d["ms_per_step"],2), d["components"]); r=d["roofline"]; print(r["frac"], r["in_step"]["frac"], r["in_step"].get("alone"), r["step_ceiling_ms"], r["step_frac"], d["cpu_baseline"]["value"], (d.get("node_masks") or d.get("reference_masks"))["ms_per_step"], d["loader_in_loop"]["ms_per_step"], d["config"]["device_memory"])
    for r in d["roofline"]["kernels"]: print("   ", r["kernel"][:70], r["avg_us"], r["frac"], r["ms_per_step"])
except Exception as e: print("bench failed", e); print(open("$O/bench_c2_n1.err").read()[-2000:])
P
grep real $O/bench_c2_n1.time
timeout 400 python bench.py --decode > $O/decode_c2.json 2> $O/decode_c2.err; tail -c 600 $O/decode_c2.json
for c in C1 C3 C5; do
  extra=""; [ $c = C1 ] && extra="--graph-leg"
  GTOS_BENCH_NO_DETAIL=1 timeout 500 python bench.py --config $c --steps 8 --warmup 3 --no-cpu-baseline --prewarm-seconds 8 $extra > $O/bench_$c.json 2> $O/bench_$c.err
  python -c "
import json
try:
    d=json.loads(open('$O/bench_$c.json').read().strip().splitlines()[-1]); print('$c', round(d['value'],1), round(d['ms_per_step'],2), d['roofline']['frac'], d['roofline']['step_frac'], d['config']['device_memory'], (d.get('node_masks') or d.get('reference_masks') or {}).get('ms_per_step'), (d.get('loader_in_loop') or {}).get('ms_per_step'), (d.get('hipgraph_replay') or {}).get('ms_per_step'))
except Exception as e: print('$c failed', e)"
done
# kernel trace of the default step
cd /tmp && export TMPDIR=/tmp && GTOS_BENCH_NO_DETAIL=1 timeout 300 rocprofv3 --kernel-trace -d $R/$O/prof -o trace -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-masks-leg --no-loader-leg --prewarm-seconds 3 > $R/$O/bench_line_under_rocprof.json 2> $R/$O/bench_rocprof.err
cd $R
DB=$(find $O/prof -name "*.db" | head -1)
python tools/rocpd_stats.py $DB $O/kernel_stats.csv > /dev/null
python tools/rocpd_stats.py $DB $O/kernel_stats_by_grid.csv --by-grid > /dev/null
python tools/rocpd_sequence.py $DB --step 3 > $O/step_sequence.txt; head -1 $O/step_sequence.txt
python tools/step_phases.py $O/step_sequence.txt > $O/step_phases.txt 2>&1; python tools/rocpd_timeline.py $DB > $O/timeline.txt 2>&1 || true
rm -rf $O/prof
head -16 $O/kernel_stats.csv | cut -c1-150
# counters: the attention kernels per operand mode, then the default step by kernel
ROUND=r6 bash tools/pmc_rel_attn.sh > $O/pmc_rel_attn.log 2>&1; tail -12 $O/pmc_rel_attn.log
ROUND=r6 MASKS=path bash tools/pmc_step.sh > $O/pmc_step.log 2>&1; head -14 gpurun_out/r6_step_pmc.txt
timeout 200 python tools/bench_gemm.py --torch --reps 5 > $O/gemm_vs_hipblaslt.txt 2>&1; grep -v amdgpu.ids $O/gemm_vs_hipblaslt.txt | grep -c "TF/s"
if [ "${PROTOCOL:-0}" = 1 ]; then
  GTOS_BENCH_NO_DETAIL=1 timeout 900 python bench.py --cpu-protocol --steps 5 --warmup 2 --no-masks-leg --no-loader-leg --prewarm-seconds 5 > $O/bench_c2_n1_cpu_protocol.json 2> $O/cpu_protocol.err
  python -c "
import json; d=json.loads(open('$O/bench_c2_n1_cpu_protocol.json').read().strip().splitlines()[-1]); print('cpu protocol', d['cpu_baseline']['value'], d['cpu_baseline']['leg'][:60], d['cpu_baseline']['sample'][:200])"
fi
