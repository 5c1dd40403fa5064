#!/bin/bash
# Round 5, call 32: parity of the default forward step kernel (three slots of activation rows + two of weight rows): bit identity over seven kernel forms, packed-path tests
O=gpurun_out/r5zh; mkdir -p $O
export PYTHONPATH=$PWD
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_full_size_c2.py -m gpu -q -x -k "ring_kernel_bit_identical or packed or relation_encoder or training_mode or fused_step" -p no:cacheprovider 2>&1 | tail -3 | tee $O/tests.txt
