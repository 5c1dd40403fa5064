#!/bin/bash
# Round 5, second GPU call: the ring-pipelined forward step (bit identity + step time), explicit operand lifetimes + bounded host run-ahead
# (reserved memory), the GRU tests again without -x.
O=gpurun_out/r5b; mkdir -p $O
export PYTHONPATH=$PWD
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -p no:cacheprovider -s \
  -k "grouped_weight or input_gradient_role or embed_packed or packed_path or training_mode_reference or fused_step_vs_golden or ring_kernel or trie_gru_equals or gru_fused_step_matches or relation_encoder" \
  > $O/tests_gru.log 2>&1
echo "gru tests rc=$? $(tail -1 $O/tests_gru.log)"
grep -E "^FAILED|^ERROR" $O/tests_gru.log | head
run() { n=$1; shift; env "$@" GTOS_BENCH_NO_DETAIL=1 timeout 300 python bench.py --no-cpu-baseline --no-loader-leg --no-masks-leg --steps 15 --warmup 3 --prewarm-seconds 6 > $O/bench_$n.json 2> $O/bench_$n.err
  python -c "
import json
d=json.loads(open('$O/bench_$n.json').read().strip().splitlines()[-1]); c=d.get('components',{}); print('$n', round(d['ms_per_step'],2), 'ms', round(d['value'],1), {k: round(v,2) for k,v in c.items() if 'relation' in k or 'gru' in k}, d['config'].get('device_memory'))" || tail -5 $O/bench_$n.err; }
run ring GTOS_X=0
run single_stage GTOS_GRU_FWD_RING=0
run ring_unbounded_runahead GTOS_MAX_STEPS_AHEAD=0
run ring_b GTOS_X=0
run single_stage_b GTOS_GRU_FWD_RING=0
run ring_ahead1 GTOS_MAX_STEPS_AHEAD=1
