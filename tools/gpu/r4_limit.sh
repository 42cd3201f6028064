cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_limit.py tests/test_gpu_parity.py tests/test_gpu_lifetime.py tests/test_gpu_operators.py -x -q -m gpu > gpurun_out/r4_limit_tests.log 2>&1 || { tail -40 gpurun_out/r4_limit_tests.log; exit 1; }
tail -3 gpurun_out/r4_limit_tests.log




