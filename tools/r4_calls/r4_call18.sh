#!/bin/bash
export PYTHONPATH=$PWD
run() { echo "== $*"; env "$@" timeout 75 python -X faulthandler tools/graph_probe.py C2 10 2>&1 | grep -v amdgpu.ids | grep "eager\|capture\|graph\|Fatal\|Error\|error\|File" | head -8; }
run GTOS_X=0
run GTOS_GRU_SIDE=0 GTOS_PROJ_SIDE=0 GTOS_BWD_SIDE=0 GTOS_GRU_L0_OVERLAP=0
