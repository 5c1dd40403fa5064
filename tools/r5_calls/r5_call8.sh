#!/bin/bash
# Round 5, eighth GPU call: ping-pong schedule of the eight-wave forward step; hipGraph leg of C1 in the default mode.
O=gpurun_out/r5h; mkdir -p $O
export PYTHONPATH=$PWD
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_zzzz_hip_graph_step.py -m gpu -q -p no:cacheprovider -k "ring_kernel or packed_path or training_mode_reference or fused_step or graphed or seed_epoch" > $O/tests_gru.log 2>&1
echo "tests rc=$? $(tail -1 $O/tests_gru.log)"; grep -E "^FAILED|^ERROR" $O/tests_gru.log | head
for nw in 8 4; do GTOS_GRU_FWD_NW=$nw timeout 200 python tools/bench_gru_step.py --reps 8 --only fwd 2>&1 | grep -v amdgpu.ids | sed "s/^/NW=$nw: /" | tee -a $O/gru_fwd_nw.txt; done
GTOS_GRU_DBG=2 timeout 200 python tools/bench_gru_step.py --reps 8 --only fwd 2>&1 | grep -v amdgpu.ids | sed "s/^/NW=8 k loop alone: /" | tee -a $O/gru_fwd_nw.txt
run() { n=$1; shift; env "$@" GTOS_BENCH_NO_DETAIL=1 timeout 300 python bench.py --no-cpu-baseline --no-loader-leg --no-masks-leg --steps 15 --warmup 3 --prewarm-seconds 6 > $O/bench_$n.json 2> $O/bench_$n.err
  python -c "
import json
d=json.loads(open('$O/bench_$n.json').read().strip().splitlines()[-1]); c=d.get('components',{}); print('$n', round(d['ms_per_step'],2), 'ms', round(d['value'],1), {k: round(v,2) for k,v in c.items() if k.endswith('_ms')}, d['config'].get('device_memory'))" || tail -5 $O/bench_$n.err; }
run nw8_pingpong GTOS_X=0
run nw4 GTOS_GRU_FWD_NW=4
run nw8_pingpong_b GTOS_X=0
run nw4_b GTOS_GRU_FWD_NW=4
timeout 300 python bench.py --config C1 --no-cpu-baseline --no-loader-leg --graph-leg --steps 30 --warmup 5 > $O/bench_C1_graph.json 2> $O/bench_C1_graph.err
python -c "
import json
d=json.loads(open('$O/bench_C1_graph.json').read().strip().splitlines()[-1]); print('C1', round(d['ms_per_step'],2), 'ms', d.get('hipgraph_replay'))" || tail -5 $O/bench_C1_graph.err
