#!/bin/bash
# Round 5, call 28: the whole GPU suite and smoke on the final library and host code (after the d4-in-place switch and the removed experiments)
O=gpurun_out/r5zd; mkdir -p $O
export PYTHONPATH=$PWD
( time timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $O/gpu_tests.log 2>&1 ) 2> $O/gpu_tests.time
tail -3 $O/gpu_tests.log | cut -c1-300; grep real $O/gpu_tests.time
timeout 300 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log | cut -c1-300
