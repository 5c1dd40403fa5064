#!/bin/bash
O=gpurun_out/r3k; mkdir -p $O
export PYTHONPATH=$PWD
snap() { echo "--- $1"; ps -eo pid,ppid,stat,rss,etime,cmd | grep -E "python|bench" | grep -v grep | cut -c1-160; rocm-smi --showmemuse 2>/dev/null | grep -E "GPU\[0\].*(VRAM|Memory)" | head -3; }
snap start
timeout 300 python -m pytest tests/test_hip_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "prefetcher" > $O/pytest_prefetcher.log 2>&1; echo "pytest prefetcher rc=$?"; tail -5 $O/pytest_prefetcher.log | cut -c1-300
snap after-pytest-prefetcher
timeout 300 python -m pytest tests/test_hip_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "fresh" > $O/pytest_fresh.log 2>&1; echo "pytest fresh rc=$?"; tail -5 $O/pytest_fresh.log | cut -c1-300
snap after-pytest-fresh
for i in 1 2 3; do
  GTOS_BENCH_NO_DETAIL=1 timeout 200 python bench.py --no-cpu-baseline --fresh-batches --workers 4 --steps 20 > $O/fresh_$i.json 2> $O/fresh_$i.err; rc=$?
  python -c "
import json
try:
    d=json.load(open('$O/fresh_$i.json')); print('fresh run $i rc=$rc', round(d['ms_per_step'],2), d['config']['loader']['consumer_wait_ms_per_step'], d['components']['graph_encoder_fwd_ms'])
except Exception as e: print('fresh run $i rc=$rc failed', e)"
  snap after-fresh-$i
done
GTOS_BENCH_NO_DETAIL=1 PYTORCH_HIP_ALLOC_CONF=roundup_power2_divisions:16 PYTORCH_CUDA_ALLOC_CONF=roundup_power2_divisions:16 timeout 200 python bench.py --no-cpu-baseline --fresh-batches --workers 4 --steps 20 > $O/fresh_round.json 2> $O/fresh_round.err; echo "rc=$?"
python -c "
import json; d=json.load(open('$O/fresh_round.json')); print('fresh roundup', round(d['ms_per_step'],2), d['config']['loader']['consumer_wait_ms_per_step'], d['components'])"
GTOS_BENCH_NO_DETAIL=1 timeout 200 python bench.py --no-cpu-baseline --steps 20 > $O/prebuilt.json 2> /dev/null; python -c "
import json; d=json.load(open('$O/prebuilt.json')); print('prebuilt', round(d['ms_per_step'],2))"
