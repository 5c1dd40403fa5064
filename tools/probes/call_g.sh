mkdir -p gpurun_out/r6g2; O=gpurun_out/r6g2
export PYTHONPATH=$PWD
( time timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $O/gpu_tests.log 2>&1 ) 2> $O/gpu_tests.time
tail -6 $O/gpu_tests.log | cut -c1-300; grep real $O/gpu_tests.time
timeout 300 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log | cut -c1-400
