#!/bin/bash
# Shows tests/test_zz_race_soak.py catching the LDS race round 5 shipped: rebuilds gru_step.o with -DGTOS_RACE_DEMO (the slot-freeing barriers of
# the pipelined k loops WITHOUT lgkmcnt(0) in front of them, as before commit 9d39564), relinks libgtos_hip.so IN THIS COPY of the tree and runs
# the GRU step soaks.  Expected: failures ("repeated launches differ from the first").  Run it on a GPU box only (gpurun's copy is scratch):
#     gpurun --timeout 900 -- 'bash tools/race_demo.sh > gpurun_out/race_demo.txt 2>&1'
set -u
cd "$(dirname "$0")/.."
export PYTHONPATH=$PWD
C=gtos_amd/csrc
cp $C/libgtos_hip.so /tmp/libgtos_hip.good.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffast-math -fno-finite-math-only -DGTOS_RACE_DEMO -c $C/gru_step.hip -o /tmp/gru_step_race.o || exit 1
OBJS=$(python -c "from gtos_amd import build; print(' '.join('$C/' + s.replace('.hip', '.o') for s in build.SOURCES if s != 'gru_step.hip'))")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/gru_step_race.o -o /tmp/libgtos_hip.race.so || exit 1
cp /tmp/libgtos_hip.race.so $C/libgtos_hip.so
echo "== soak with the pre-fix waits (expected: FAILURES) =="
GTOS_SOAK_REPS=${GTOS_SOAK_REPS:-60} timeout 600 python -m pytest tests/test_zz_race_soak.py -q --tb=line -p no:cacheprovider -k "gru_step" 2>&1 | tail -25
cp /tmp/libgtos_hip.good.so $C/libgtos_hip.so
echo "== soak with the product library (expected: all passed) =="
timeout 600 python -m pytest tests/test_zz_race_soak.py -q --tb=line -p no:cacheprovider -k "gru_step" 2>&1 | tail -5
