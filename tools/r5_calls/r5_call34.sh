export PYTHONPATH=$PWD
mkdir -p gpurun_out/r5zj
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "input_gradient_role" -p no:cacheprovider --tb=long 2>&1 | tail -60 | cut -c1-250 > gpurun_out/r5zj/fail.txt
GTOS_GRU_BWD_DBG=3 timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "input_gradient_role" -p no:cacheprovider --tb=short 2>&1 | tail -3 >> gpurun_out/r5zj/fail.txt
