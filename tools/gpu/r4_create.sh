cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests/test_gpu_single_pass.py tests/test_gpu_parity.py tests/test_gpu_threads.py tests/test_gpu_fault_injection.py -x -q -m gpu > gpurun_out/r4_create_tests.log 2>&1 || { tail -30 gpurun_out/r4_create_tests.log; exit 1; }
tail -3 gpurun_out/r4_create_tests.log
python tools/first_run.py 2>&1 | tee gpurun_out/r4_first_run.log
