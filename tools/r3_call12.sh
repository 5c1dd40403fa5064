#!/bin/bash
O=gpurun_out/r3l; mkdir -p $O
export PYTHONPATH=$PWD
snap() { echo "--- $1: $(rocm-smi --showmemuse 2>/dev/null | grep 'VRAM%' | head -1)"; for p in /proc/[0-9]*; do c=$(tr '\0' ' ' < $p/cmdline 2>/dev/null | cut -c1-100); case "$c" in *python*|*bench*) echo "   pid $(basename $p) ppid $(awk '{print $4}' $p/stat 2>/dev/null) state $(awk '{print $3}' $p/stat 2>/dev/null): $c";; esac; done; }
snap start
timeout 300 python -m pytest tests/test_hip_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "prefetcher or fresh" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log | cut -c1-200
snap after-pytest
for i in 1 2 3 4; do
  GTOS_BENCH_VERBOSE=1 GTOS_BENCH_NO_DETAIL=1 timeout 200 python bench.py --no-cpu-baseline --fresh-batches --workers 4 --steps 20 > $O/fresh_$i.json 2> $O/fresh_$i.err; rc=$?
  python -c "
import json
try:
    d=json.load(open('$O/fresh_$i.json')); print('fresh run $i rc=$rc', round(d['ms_per_step'],2), d['config']['loader'])
except Exception as e: print('fresh run $i rc=$rc failed', e)"
  snap after-fresh-$i
  sleep 3
  snap after-fresh-$i-plus3s
done
for i in 1 2; do
GTOS_BENCH_NO_ROUNDUP=1 GTOS_BENCH_VERBOSE=1 GTOS_BENCH_NO_DETAIL=1 timeout 200 python bench.py --no-cpu-baseline --fresh-batches --workers 4 --steps 20 > $O/fresh_noround_$i.json 2> $O/fresh_noround_$i.err; echo "rc=$?"
python -c "
import json; d=json.load(open('$O/fresh_noround_$i.json')); print('fresh no-roundup $i', round(d['ms_per_step'],2), d['config']['loader'])"
done
GTOS_BENCH_NO_DETAIL=1 timeout 200 python bench.py --no-cpu-baseline --steps 20 > $O/prebuilt.json 2> /dev/null; python -c "
import json; d=json.load(open('$O/prebuilt.json')); print('prebuilt', round(d['ms_per_step'],2))"
