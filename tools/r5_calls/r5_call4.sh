#!/bin/bash
# Round 5, fourth GPU call: the whole GPU suite on the round's code so far; where the forward step kernel's time goes (k loop alone /
# cell alone); attention kernels with the type ids staged in LDS and the bank kernel with two pairs in flight; the step.
O=gpurun_out/r5d; mkdir -p $O
export PYTHONPATH=$PWD
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/gpu_tests.log 2>&1
echo "gpu suite rc=$? $(tail -1 $O/gpu_tests.log)"; grep -E "^FAILED|^ERROR" $O/gpu_tests.log | head
for dbg in 0 1 2; do GTOS_GRU_DBG=$dbg timeout 200 python tools/bench_gru_step.py --reps 8 --only fwd 2>&1 | grep -v amdgpu.ids | sed "s/^/dbg=$dbg (0 full, 1 no k loop, 2 no cell): /" | tee -a $O/gru_fwd_parts.txt; done
timeout 300 python tools/bench_rel_attn.py --reps 20 > $O/rel_attn_np2.txt 2>&1; grep -v amdgpu.ids $O/rel_attn_np2.txt | tail -30
GTOS_BANK_NP=4 timeout 300 python tools/bench_rel_attn.py --reps 20 --mode factored > $O/rel_attn_np4.txt 2>&1; grep -v amdgpu.ids $O/rel_attn_np4.txt | tail -12
run() { n=$1; shift; env "$@" GTOS_BENCH_NO_DETAIL=1 timeout 300 python bench.py --no-cpu-baseline --no-loader-leg --no-masks-leg --steps 15 --warmup 3 --prewarm-seconds 6 > $O/bench_$n.json 2> $O/bench_$n.err
  python -c "
import json
d=json.loads(open('$O/bench_$n.json').read().strip().splitlines()[-1]); c=d.get('components',{}); print('$n', round(d['ms_per_step'],2), 'ms', round(d['value'],1), {k: round(v,2) for k,v in c.items() if k.endswith('_ms')}, d['config'].get('device_memory'))" || tail -5 $O/bench_$n.err; }
run default GTOS_X=0
run bank_np4 GTOS_BANK_NP=4
run default_b GTOS_X=0
