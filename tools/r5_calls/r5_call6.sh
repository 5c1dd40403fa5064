#!/bin/bash
# Round 5, sixth GPU call: fast tanh in the bf16 step kernels (the forward cell was VALU-bound), measuring switches as template
# instantiations (the runtime switch had pushed the production kernel into spills).
O=gpurun_out/r5f; mkdir -p $O
export PYTHONPATH=$PWD
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_pathtrie.py -m gpu -q -p no:cacheprovider \
  -k "gru or relation_encoder or packed or trie" > $O/tests_gru.log 2>&1
echo "gru tests rc=$? $(tail -1 $O/tests_gru.log)"; grep -E "^FAILED|^ERROR" $O/tests_gru.log | head
for dbg in 0 1 2; do GTOS_GRU_DBG=$dbg timeout 200 python tools/bench_gru_step.py --reps 8 --only fwd 2>&1 | grep -v amdgpu.ids | sed "s/^/dbg=$dbg (0 full, 1 no k loop, 2 no cell): /" | tee -a $O/gru_fwd_parts.txt; done
run() { n=$1; shift; env "$@" GTOS_BENCH_NO_DETAIL=1 timeout 300 python bench.py --no-cpu-baseline --no-loader-leg --no-masks-leg --steps 15 --warmup 3 --prewarm-seconds 6 > $O/bench_$n.json 2> $O/bench_$n.err
  python -c "
import json
d=json.loads(open('$O/bench_$n.json').read().strip().splitlines()[-1]); c=d.get('components',{}); print('$n', round(d['ms_per_step'],2), 'ms', round(d['value'],1), {k: round(v,2) for k,v in c.items() if k.endswith('_ms')}, d['config'].get('device_memory'))" || tail -5 $O/bench_$n.err; }
run default GTOS_X=0
run single_stage GTOS_GRU_FWD_RING=0
run round4_path GTOS_RELENC_PACKED=0
run default_b GTOS_X=0
run node_masks GTOS_RELENC_MASKS=node
