#!/bin/bash
# Round 5, fifth GPU call: forward directions on two streams (A/B), the full-size C2 tests incl. the new training-mode leg.
O=gpurun_out/r5e; mkdir -p $O
export PYTHONPATH=$PWD
run() { n=$1; shift; env "$@" GTOS_BENCH_NO_DETAIL=1 timeout 300 python bench.py --no-cpu-baseline --no-loader-leg --no-masks-leg --steps 15 --warmup 3 --prewarm-seconds 6 > $O/bench_$n.json 2> $O/bench_$n.err
  python -c "
import json
d=json.loads(open('$O/bench_$n.json').read().strip().splitlines()[-1]); c=d.get('components',{}); print('$n', round(d['ms_per_step'],2), 'ms', round(d['value'],1), {k: round(v,2) for k,v in c.items() if k.endswith('_ms')}, d['config'].get('device_memory'))" || tail -5 $O/bench_$n.err; }
run fwd_two_streams GTOS_X=0
run fwd_one_stream GTOS_GRU_FWD_OVERLAP=0
run fwd_two_streams_b GTOS_X=0
run fwd_one_stream_b GTOS_GRU_FWD_OVERLAP=0
timeout 1200 python -m pytest tests/test_full_size_c2.py -m gpu -q -s -p no:cacheprovider > $O/full_size.log 2>&1
echo "full-size rc=$? $(tail -1 $O/full_size.log)"; grep -E "^C2 |^FAILED|^ERROR|Error" $O/full_size.log | cut -c1-400 | head -20
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -p no:cacheprovider -k "packed_path or training_mode_reference or fused_step_vs_golden or beam or decode" > $O/tests_sel.log 2>&1
echo "selected rc=$? $(tail -1 $O/tests_sel.log)"
timeout 600 python -m pytest tests/test_beam_and_vocab.py -m gpu -q -p no:cacheprovider > $O/tests_beam.log 2>&1
echo "beam rc=$? $(tail -1 $O/tests_beam.log)"
