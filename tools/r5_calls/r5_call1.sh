#!/bin/bash
# Round 5, first GPU call: the packed-path RelationEncoder -- parity tests of the new kernels and of the module, then the training step
# with each of its parts switched off in turn (same box, alternating legs).
O=gpurun_out/r5a; mkdir -p $O
export PYTHONPATH=$PWD
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -x -p no:cacheprovider -s \
  -k "grouped_weight or input_gradient_role or embed_packed or packed_path or training_mode_reference or fused_step_vs_golden or weight_gradient_long_k or trie_gru_equals or gru_fused_step_matches or relation_encoder" \
  > $O/tests_gru.log 2>&1
echo "gru tests rc=$? $(tail -1 $O/tests_gru.log)"
grep -E "^MEASURED|measured" $O/tests_gru.log | tail -40
run() { n=$1; shift; env "$@" GTOS_BENCH_NO_DETAIL=1 timeout 300 python bench.py --no-cpu-baseline --no-loader-leg --no-masks-leg --steps 15 --warmup 3 --prewarm-seconds 6 > $O/bench_$n.json 2> $O/bench_$n.err
  python -c "
import json
d=json.loads(open('$O/bench_$n.json').read().strip().splitlines()[-1]); c=d.get('components',{}); print('$n', round(d['ms_per_step'],2), 'ms', round(d['value'],1), {k: round(v,2) for k,v in c.items() if 'relation' in k or 'gru' in k}, d['config'].get('device_memory'))" || tail -5 $O/bench_$n.err; }
run round4_path GTOS_RELENC_PACKED=0
run packed GTOS_X=0
run packed_no_dinp_fusion GTOS_GRU_FUSE_DINP=0
run packed_no_dw_merge GTOS_GRU_MERGE_DW=0
run round4_path_b GTOS_RELENC_PACKED=0
run packed_b GTOS_X=0
